#!/bin/bash
# round 5, session 8: same-box A/B of the two tile forms of the split-bf16 body
# (segment-aligned + Q row in registers vs 16 consecutive edges + per-lane Q)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name tune args
  echo "== $1"
  PGNN_TUNE=$2 timeout 200 python tools/bf16x3_bench.py $3 2>&1 | grep -E "split-bf16|max \|bf16x3|E1 "
}
run "segment-aligned" b16_seg=1 ""
run "16 consecutive edges" b16_seg=0 ""
run "segment-aligned" b16_seg=1 ""
run "16 consecutive edges" b16_seg=0 ""
run "ped: segment-aligned" b16_seg=1 "--preset ped_dense --config ped_cyl_auto_T3"
run "ped: 16 consecutive edges" b16_seg=0 "--preset ped_dense --config ped_cyl_auto_T3"
run "car: segment-aligned" b16_seg=1 "--preset car"
run "car: 16 consecutive edges" b16_seg=0 "--preset car"
PGNN_TUNE=b16_seg=0 timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
PGNN_TUNE=b16_seg=1 timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
