#!/bin/bash
# Round-3 GPU session 5: kernel trace of the steady-state frame pipeline.
# usage: r03_s5.sh <tag> [pipe_run.py arguments]
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s5
mkdir -p $OUT
export TMPDIR=/tmp
tag=$1; shift
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_$tag -o run -- python $ROOT/tools/pipe_run.py --frames 16 "$@" > $OUT/run_$tag.log 2>&1)
grep "frames/s" $OUT/run_$tag.log
db=$(find $OUT/prof_$tag -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 30 --out $OUT/trace_$tag.txt
rm -rf $OUT/prof_$tag
