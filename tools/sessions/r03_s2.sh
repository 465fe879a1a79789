#!/bin/bash
# Round-3 GPU session 2: training-step work -- the training tests, the
# training bench line, a per-launch trace of one step.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s2
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
( timeout 600 python bench.py --train --steps 24 --warmup 8 --frames 4 ) > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -c 900 $OUT/bench_train.json; tail -3 $OUT/bench_train.err
if [ "${1:-}" != "notrace" ]; then
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_train -o run -- python $ROOT/bench.py --train --steps 4 --warmup 3 --frames 4 > $OUT/prof_train.log 2>&1)
db=$(find $OUT/prof_train -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 12 --out $OUT/train_trace.txt
rm -rf $OUT/prof_train
grep -A45 "^window" $OUT/train_trace.txt
fi
