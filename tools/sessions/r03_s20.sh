#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s20
timeout 300 python tools/builder_cost.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s20/builder_cost.log
