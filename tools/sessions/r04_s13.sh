#!/bin/bash
# round 4, session 13: what the split-bf16 kernel waits for -- ablations
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "" NOSPLIT NOGATHER NOMFMA; do
  echo "== variant ${v:-tree}"
  if [ -z "$v" ]; then timeout 300 python tools/bf16x3_bench.py; else PGNN_LIB=$PWD/ab/libb16$v.so timeout 300 python tools/bf16x3_bench.py; fi
done 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s13_b16_abl.txt
cat gpurun_out/r04_s13_b16_abl.txt
