#!/bin/bash
# graph build alone / beside the GNN kernels, per library variant and CU split
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s6
mkdir -p $OUT
for spec in "$@"; do
  lib=${spec%%:*}; cus=${spec#*:}; [ "$cus" = "$spec" ] && cus=0
  if [ "$lib" = "default" ]; then unset PGNN_LIB; else export PGNN_LIB=$ROOT/ab/lib$lib.so; fi
  echo "== $lib graph_cus=$cus"
  timeout 300 python tools/corun.py --graph-cus $cus 2>&1 | grep -v amdgpu.ids | tee $OUT/corun_${lib}_$cus.log
done
