#!/bin/bash
# round 5, session 4: what one more instruction costs beside a bf16 MFMA stream
# (tools/micro/mfma_mix.hip), and the hybrid split of the interleaved body
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 ./ab/mfma_mix | tee gpurun_out/r05_s4_mfma_mix.txt
run() {  # name lib
  echo "== $1"
  PGNN_LIB=$2 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "split-bf16|max \|bf16x3"
}
run "hybrid split" $PWD/ab/libhyb.so
PGNN_LIB=$PWD/ab/libhyb.so timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -3
