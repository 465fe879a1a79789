#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s23 ab
[ -f ab/libprobe.so ] || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 tools/micro/corun_probe.hip -o ab/libprobe.so
timeout 300 python tools/corun_mlp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s23/corun_mlp.log
