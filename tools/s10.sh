#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pooling or epilogue or end_to_end or full_size" 2>&1 | tail -4
for d in 1024 0 1024 0; do
  echo "== mlp_debug $d"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 64 --tune mlp_debug=$d 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f pool_us %.1f pool_frac %.3f' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))"
done
