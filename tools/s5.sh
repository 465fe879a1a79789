#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined or epilogue or end_to_end" 2>&1 | tail -3
for g in 0 8 16 32 0 8 16 32; do
  echo "== graph_cus $g"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 96 --graph-cus $g 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f' % b['value'])"
done
