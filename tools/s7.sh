#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined or epilogue or end_to_end or pooling_layer or auto_center_layer" 2>&1 | tail -3
for pct in 0 12 25 0 12 25 6; do
  echo "== pool_pct $pct"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 96 --lookahead 0 --tune mlp_pool_pct=$pct 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f pool_us %.1f' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_pool']['avg_launch_us']))"
done
