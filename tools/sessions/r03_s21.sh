#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s21
timeout 300 python tools/graph_capture_probe.py 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/s21/capture.log
