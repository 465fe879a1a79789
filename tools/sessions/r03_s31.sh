#!/bin/bash
# radius query: two candidate chunks per round + reciprocal cell function
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py tests/test_gpu_f64.py -m gpu -x -q -k "radius or graph or deferred or overlap or stages or multi_level or f64 or frame" 2>&1 | tail -2
rm -rf $OUT/prof_build
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_build -o run -- python $ROOT/tools/build_trace.py --stream > $OUT/prof_build.log 2>&1)
grep overlap $OUT/prof_build.log | tail -3
db=$(find $OUT/prof_build -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 1.0 --out $OUT/build_trace.txt
rm -rf $OUT/prof_build
grep -n "radius_query\|window" $OUT/build_trace.txt | cut -c1-110
