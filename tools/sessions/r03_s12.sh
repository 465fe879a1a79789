#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s12
timeout 300 python tools/${1:-hog_probe}.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s12/$1.log
