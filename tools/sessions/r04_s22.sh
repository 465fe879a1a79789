#!/bin/bash
# round 4, session 22: branch-free K loop in the 4-wave LDS-tile kernels (ped pooling)
cd "$GRAFT_REPO_ROOT"
for v in "" kbf kbf_pf4 ""; do
  if [ -z "$v" ]; then timeout 300 python tools/ped_pool_bench.py; else PGNN_LIB=$PWD/ab/lib$v.so timeout 300 python tools/ped_pool_bench.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s22_ped_pool.txt
