#!/bin/bash
# round 5, session 7: segment-aligned tiles, Q row in registers (DPP row_newbcast)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -8
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "split-bf16|max \|bf16x3"
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "tree (segment-aligned)" $T
run "dp4" $PWD/ab/libdp4.so
run "abl 32 no P requests" $PWD/ab/libabl32.so
run "abl 36 no P requests, fragments once" $PWD/ab/libabl36.so
run "abl 16 one wave per SIMD" $PWD/ab/libabl16.so
run "tree again" $T
bash tools/pmc_b16.sh $PWD/gpurun_out/r05_s7_pmc_b16.txt 3 | grep b16x3
