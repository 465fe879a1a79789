#!/bin/bash
# round 5, session 11: the two-part fp16 edge kernel (edge_ws_f16.h): first run
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/bf16x3_bench.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 300 python tools/bf16x3_bench.py --preset ped_dense --config ped_cyl_auto_T3 2>&1 | grep -v amdgpu.ids | tail -6
timeout 1500 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -25
