#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
  -k "weights_stationary or end_to_end or full_size_logits" 2>&1 | tail -5
echo "=== timeline default"; timeout 200 python tools/ws_timeline.py 2>&1 | grep -v amdgpu.ids
echo "=== timeline pool 0"; timeout 200 python tools/ws_timeline.py --tune=ws_pool_pct=0 2>&1 | grep -v amdgpu.ids
for t in mlp_debug=2048 mlp_debug=0 ws_prio=0 ws_pool_pct=0 "ws_pool_pct=30 --tune ws_chunk=4" mlp_debug=0; do
  echo "== $t"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 64 --tune $t 2>gpurun_out/s16_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us']))" \
    || tail -5 gpurun_out/s16_err.log
done
