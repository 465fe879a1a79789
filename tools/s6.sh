#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined" 2>&1 | tail -3
for la in 0 1 2 3 0 2; do
  for g in 0 16; do
  echo "== lookahead $la graph_cus $g"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 96 --graph-cus $g --lookahead $la 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f' % b['value'])"
  done
done
