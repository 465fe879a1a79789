#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s9
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o run -- python $ROOT/tools/corun.py --graph-cus ${1:-8} > $OUT/run.log 2>&1)
grep -v "rocprof\|^W2026\|^E2026\|amdgpu.ids" $OUT/run.log | tail -4
db=$(find $OUT/prof -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 40 --out $OUT/trace.txt
rm -rf $OUT/prof
