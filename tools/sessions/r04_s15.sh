#!/bin/bash
# round 4, session 18: split-bf16 edge kernel, constant-offset gather addressing
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 300 python tools/bf16x3_bench.py; timeout 300 python tools/bf16x3_bench.py --preset ped_dense --config ped_cyl_auto_T3 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_s15_bf16.txt
cat gpurun_out/r04_s15_bf16.txt
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu -s 2>&1 | tail -9
