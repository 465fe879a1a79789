#!/bin/bash
# round 5, session 2: what bounds the interleaved split-bf16 edge body --
# timing ablations (wrong results: rows from one address / one part / fragments
# read once / one term, and pairs of them) and the SQ counters of the tree's kernel.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep "split-bf16"
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "tree (interleaved, rn)" $T
run "abl 1 rows from one address" $PWD/ab/libabl1.so
run "abl 2 one part (no split)" $PWD/ab/libabl2.so
run "abl 4 fragments read once" $PWD/ab/libabl4.so
run "abl 8 one term of six" $PWD/ab/libabl8.so
run "abl 3 = 1+2" $PWD/ab/libabl3.so
run "abl 5 = 1+4" $PWD/ab/libabl5.so
run "abl 6 = 2+4" $PWD/ab/libabl6.so
run "abl 7 = 1+2+4" $PWD/ab/libabl7.so
run "tree again" $T
bash tools/pmc_b16.sh $PWD/gpurun_out/r05_s2_pmc_b16.txt 5
