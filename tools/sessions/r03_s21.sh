#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s21
for i in 1 2 3; do
timeout 300 python tools/graph_capture_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tail -6
done
