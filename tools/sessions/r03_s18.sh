#!/bin/bash
cd "$(dirname "$0")/../.."
for extra in "$@"; do
timeout 300 python bench.py --config ped_cyl_auto_T3 --no-cpu-baseline --no-live-pmc --no-roofline --steps 32 $extra 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('ped [$extra] fps %.1f' % d['value'])"
done
