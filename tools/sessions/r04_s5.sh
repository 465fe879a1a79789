#!/bin/bash
# round 4, session 5: K-row kernels standalone, prefetch depth 4 / 6 / 8 now
# that the K loop no longer drains at its header
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "" pf6 pf8 ""; do
  if [ -z "$v" ]; then python tools/krow_bench.py; else PGNN_LIB=$PWD/ab/lib$v.so python tools/krow_bench.py; fi
done > gpurun_out/r04_s5_krow.txt 2>&1
cat gpurun_out/r04_s5_krow.txt
