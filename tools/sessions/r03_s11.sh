#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s11
timeout 300 python tools/gnn_only.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s11/gnn_only.log
