#!/bin/bash
# kd-tree build time of every ab/libkd_*.so variant vs the tree's library
# (equality with sklearn's KDTree.get_arrays() checked for each)
cd "$(dirname "$0")/.."
for lib in point-gnn_amd/libpointgnn_hip.so ab/libkd_*.so; do
  echo "== $lib"
  PGNN_LIB=$PWD/$lib timeout 300 python tools/kd_check.py n9000 car car_600k ped_dense 2>&1 | grep -v Warning
done
