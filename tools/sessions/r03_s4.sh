#!/bin/bash
# Round-3 GPU session 4: graph-build kernels beside the persistent MFMA kernels
# (wave priority, small-footprint kd-tree build): same-box A/B of the bench.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s4
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset PGNN_LIB; else export PGNN_LIB=$ROOT/ab/lib$lib.so; fi
  echo "== $lib"
  ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kdtree or keypoints or pipelined" ) > $OUT/pytest_$lib.log 2>&1
  tail -2 $OUT/pytest_$lib.log
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-secondary --steps 64 2>$OUT/err_$lib.log > $OUT/bench_${lib}_$rep.json
  python - <<PY
import json
try:
    b=json.load(open("$OUT/bench_${lib}_$rep.json")); c=b["config"]
    print("$lib rep $rep: frames/s %.1f  edge_us %.1f pool_us %.1f gen_graph_ms %.3f latency %s" % (b["value"], b["roofline_mfma"]["avg_launch_us"], b["roofline_pool"]["avg_launch_us"], c["phase_ms_frame_seed0"]["gen graph"], {k: round(v,3) for k,v in c["latency_ms_frame_seed0"].items()}))
except Exception as e:
    print("$lib failed", e); print(open("$OUT/err_$lib.log").read()[-800:])
PY
  done
done
