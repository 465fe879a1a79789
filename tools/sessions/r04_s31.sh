#!/bin/bash
# round 4, session 31: the fused edge kernel at the training step's shape --
# standalone timings of the three entries, and the wave timeline
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/train_edge_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s31_train_edge.txt
timeout 300 python tools/ws_timeline.py --train-graph --balance 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_s31_ws_timeline_train.txt
