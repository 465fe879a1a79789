#!/bin/bash
# per-kernel durations of one sequential frame (no overlap)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s15
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --no-pipeline --frames 1 --steps 12 --warmup 3 "$@" > $OUT/run.log 2>&1)
db=$(find $OUT/prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" $OUT/stats > /dev/null
rm -rf $OUT/prof
head -45 $OUT/stats.md | cut -c1-120
