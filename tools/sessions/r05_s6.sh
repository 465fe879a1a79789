#!/bin/bash
# round 5, session 6: DPP instead of ds_bpermute at the tile boundary; timing
# ablations: no row requests (32), no segmented max (64), fragments read once (4)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "split-bf16"
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "tree" $T
run "abl 32 no row requests" $PWD/ab/libabl32.so
run "abl 64 no segmented max" $PWD/ab/libabl64.so
run "abl 4 fragments once" $PWD/ab/libabl4.so
run "abl 36" $PWD/ab/libabl36.so
run "abl 96" $PWD/ab/libabl96.so
run "abl 100 = 4 + 32 + 64" $PWD/ab/libabl100.so
run "tree" $T
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -2
