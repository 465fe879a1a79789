#!/bin/bash
# round 4, session 32: more than one pooling level (pgnn_voxel_keypoints_*_from)
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_multilevel.py -q -m gpu -s 2>&1 | tail -40 | tee gpurun_out/r04_s32_tests.txt
