#!/bin/bash
# round 4, session 16: split-bf16 edge kernel, second wave of each SIMD started late
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "" b16st0 b16st24 b16st96 b16st192; do
  echo "== ${v:-tree (stagger 48)}"
  if [ -z "$v" ]; then timeout 300 python tools/bf16x3_bench.py; else PGNN_LIB=$PWD/ab/lib$v.so timeout 300 python tools/bf16x3_bench.py; fi
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s16_bf16.txt
cat gpurun_out/r04_s16_bf16.txt
