#!/bin/bash
# round 4, session 23: split-bf16 kernel, residuals by v_dot2_f32_bf16 (exactness + time)
cd "$GRAFT_REPO_ROOT"
for v in "" b16dot2 ""; do
  if [ -z "$v" ]; then timeout 300 python tools/bf16x3_bench.py; else PGNN_LIB=$PWD/ab/lib$v.so timeout 300 python tools/bf16x3_bench.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s23_dot2.txt
