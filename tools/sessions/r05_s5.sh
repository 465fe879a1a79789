#!/bin/bash
# round 5, session 5: interleaved split-bf16 body without packed-fp32 ops,
# stage-major split, row requests spread over the block
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "split-bf16|max \|bf16x3"
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "tree (hybrid split, loads every 6, dp3 dq2)" $T
run "all parts rounded (68 VALU)" $PWD/ab/librn.so
run "all parts cut" $PWD/ab/libtr.so
run "row requests in a burst" $PWD/ab/lible1.so
run "dp4 dq3" $PWD/ab/libdp4.so
run "one wave per SIMD" $PWD/ab/libabl16.so
run "tree again" $T
bash tools/pmc_b16.sh $PWD/gpurun_out/r05_s5_pmc_b16.txt 3 | grep b16x3
