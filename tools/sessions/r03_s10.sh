#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s10
timeout 300 python tools/cu_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s10/cu_probe.log
