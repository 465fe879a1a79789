#!/bin/bash
# capped builder grids: parity in both modes, then graph build alone / beside
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s13
rm -rf $OUT; mkdir -p $OUT
CAP="graph_max_wgs=8,graph_lds_pad=32768,ws_reserve=8"
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f64.py tests/test_gpu_deferred.py -m gpu -x -q ) > $OUT/pytest_default.log 2>&1
tail -3 $OUT/pytest_default.log
( PGNN_TUNE=$CAP timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f64.py tests/test_gpu_deferred.py -m gpu -x -q ) > $OUT/pytest_capped.log 2>&1
tail -3 $OUT/pytest_capped.log
echo "== corun default"; timeout 300 python tools/corun.py 2>&1 | grep -v amdgpu.ids
echo "== corun capped"; timeout 300 python tools/corun.py --tune graph_max_wgs=8 --tune graph_lds_pad=32768 --tune ws_reserve=8 2>&1 | grep -v amdgpu.ids
echo "== corun capped 16"; timeout 300 python tools/corun.py --tune graph_max_wgs=16 --tune graph_lds_pad=32768 --tune ws_reserve=16 2>&1 | grep -v amdgpu.ids
bash tools/sessions/r03_s8.sh "" "--compute-streams 1 --tune graph_max_wgs=8 --tune graph_lds_pad=32768 --tune ws_reserve=8" "--compute-streams 2 --tune graph_max_wgs=8 --tune graph_lds_pad=32768 --tune ws_reserve=8" "--compute-streams 1 --tune graph_max_wgs=16 --tune graph_lds_pad=32768 --tune ws_reserve=16"
