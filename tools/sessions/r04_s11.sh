#!/bin/bash
# round 4, session 11: the split-bf16 edge kernel -- first run
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python tools/bf16x3_bench.py > gpurun_out/r04_s11_bf16.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_s11_bf16.txt | tail -8
timeout 900 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu -s 2>&1 | tail -15 > gpurun_out/r04_s11_tests.log
cat gpurun_out/r04_s11_tests.log
