#!/bin/bash
# probe: step latencies of small dependent chains beside the GNN kernels
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s7 ab
[ -f ab/libprobe.so ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 tools/micro/corun_probe.hip -o ab/libprobe.so
timeout 300 python tools/corun_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s7/probe.log
