#!/bin/bash
# Round-3 GPU session 1: GPU suite on the float64 / raster / sched changes,
# the new default bench line, and a per-launch trace of the training step.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s1
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_train -o run -- python $ROOT/bench.py --train --steps 4 --warmup 3 --no-cpu-baseline > $OUT/prof_train.log 2>&1)
db=$(find $OUT/prof_train -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 14 --out $OUT/train_trace.txt
python tools/prof_summary.py "$db" $OUT/train_kernel_stats > /dev/null
rm -rf $OUT/prof_train
tail -c 600 $OUT/prof_train.log
tail -45 $OUT/train_trace.txt
