#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s17
timeout 300 python tools/frame_streams.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s17/frame_streams.log
