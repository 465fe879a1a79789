#!/bin/bash
# kernel timeline of the graph build alone: in order vs overlapped
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_parity.py -m gpu -x -q -k "deferred or overlap or stages or center or kdtree or captured" 2>&1 | tail -2
timeout 120 python tools/build_trace.py --stream 2>&1 | grep "overlap\|main"
rm -rf $OUT/prof_build
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_build -o run -- python $ROOT/tools/build_trace.py --stream > $OUT/prof_build.log 2>&1)
db=$(find $OUT/prof_build -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 1.0 --out $OUT/build_trace.txt
rm -rf $OUT/prof_build
head -75 $OUT/build_trace.txt | cut -c1-110
