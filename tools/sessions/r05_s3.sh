#!/bin/bash
# round 5, session 3: interleaved split-bf16 body -- hybrid split (first part
# rounded, the rest cut), request distances, static priority of the younger
# waves, one wave per SIMD (timing ablation)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "split-bf16|max \|bf16x3"
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "tree (interleaved, rn, dp3 dq2)" $T
run "hybrid split" $PWD/ab/libhyb.so
run "rn dp4 dq3" $PWD/ab/libdp4.so
run "hybrid dp4 dq3" $PWD/ab/libhyb_dp4.so
run "younger waves at prio 1" $PWD/ab/libyp.so
run "abl 16: one wave per SIMD" $PWD/ab/libabl16.so
run "abl 24: one wave per SIMD, one term" $PWD/ab/libabl24.so
run "abl 23: one wave per SIMD, 1+2+4" $PWD/ab/libabl23.so
run "tree again" $T
PGNN_LIB=$PWD/ab/libhyb.so timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
