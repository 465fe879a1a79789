#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s22
timeout 300 python tools/graph_replay_streams.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/s22/replay.log
