#!/bin/bash
cd "$(dirname "$0")/.."
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 96 "$@" 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f' % b['value'])"; }
for q in 4 8 16; do
  for dbg in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$q graph_debug=$dbg"
    GPU_MAX_HW_QUEUES=$q run --lookahead 0 --tune graph_debug=$dbg
  done
done
echo "== queues 8 lookahead 2"; GPU_MAX_HW_QUEUES=8 run --lookahead 2
echo "== queues 8 compute-streams 1"; GPU_MAX_HW_QUEUES=8 run --lookahead 0 --compute-streams 1
echo "== queues 4 compute-streams 1"; run --lookahead 0 --compute-streams 1
echo "== queues 8 compute-streams 3"; GPU_MAX_HW_QUEUES=8 run --lookahead 0 --compute-streams 3
