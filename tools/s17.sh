#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
  -k "weights_stationary or pooling or end_to_end or full_size_logits or pipelined" 2>&1 | tail -8
echo "=== timeline default + dump"; timeout 200 python tools/ws_timeline.py --dump 2>&1 | grep -v amdgpu.ids
for t in mlp_debug=0 mlp_debug=8192 mlp_debug=10240 ws_pool_pct=0 mlp_debug=0; do
  echo "== $t"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 64 --tune $t 2>gpurun_out/s17_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f)' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))" \
    || tail -5 gpurun_out/s17_err.log
done
