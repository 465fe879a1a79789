#!/bin/bash
# round 5, session 1: the interleaved split-bf16 edge body (edge_ws_bf16.h) --
# correctness, then same-box timing of: the two-phase body at the three
# priority modes, the interleaved body with truncating / rounding splits and
# other request distances.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bf16x3.py -x -q -m gpu 2>&1 | tail -5
run() {  # name lib tune
  echo "== $1"
  PGNN_LIB=$2 PGNN_TUNE=$3 timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -v amdgpu.ids | tail -5
}
T=$PWD/point-gnn_amd/libpointgnn_hip.so
run "old body prio=1 (r04)"   $PWD/ab/libold.so ws_prio=1
run "old body prio=0"         $PWD/ab/libold.so ws_prio=0
run "old body prio=2 (MFMA phases raised)" $PWD/ab/libold.so ws_prio=2
run "interleaved trunc dp3 dq2 (tree)" $T ""
run "interleaved rn"          $PWD/ab/libil_rn.so ""
run "interleaved dq3"         $PWD/ab/libil_dq3.so ""
run "interleaved dp4 dq3"     $PWD/ab/libil_dp4.so ""
run "interleaved (tree) again" $T ""
