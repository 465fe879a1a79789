#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s27
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python $ROOT/bench.py --train --steps 8 --warmup 4 --frames 4 > $OUT/run.log 2>&1)
db=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" $OUT/train_stats > /dev/null
rm -rf $OUT/prof
head -24 $OUT/train_stats.md | cut -c1-105
