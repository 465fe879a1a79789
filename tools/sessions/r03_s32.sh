#!/bin/bash
# voxel_leader / voxel_random_pick: four slots per round trip
cd "$(dirname "$0")/../.."
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f64.py tests/test_gpu_deferred.py -m gpu -x -q -k "keypoint or center or random or voxel or kdtree or multi_level or f64 or deferred_graph" 2>&1 | tail -2
timeout 120 python tools/build_trace.py --stream 2>&1 | grep "overlap" | tail -8
