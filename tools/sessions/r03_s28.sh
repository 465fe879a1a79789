#!/bin/bash
cd "$(dirname "$0")/../.."
for t in 768 512 256 1024; do
timeout 300 python bench.py --train --steps 24 --warmup 8 --frames 4 --tune wgrad_wg_target=$t 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('wgrad_wg_target $t: train fps %.1f ms %.3f' % (d['value'], d['ms_per_step']))"
done
