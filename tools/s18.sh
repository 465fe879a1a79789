#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
  -k "weights_stationary or end_to_end or full_size_logits or pipelined" 2>&1 | tail -8
echo "=== timeline pipe=1"; timeout 200 python tools/ws_timeline.py --balance --dump 2>&1 | grep -v amdgpu.ids
echo "=== timeline pipe=0"; timeout 200 python tools/ws_timeline.py --tune=ws_pipe=0 --balance 2>&1 | grep -v amdgpu.ids
for t in ws_pipe=0 ws_pipe=1 ws_pipe=0 ws_pipe=1; do
  echo "== $t"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 64 --tune $t 2>gpurun_out/s18_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f)' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))" \
    || tail -5 gpurun_out/s18_err.log
done
