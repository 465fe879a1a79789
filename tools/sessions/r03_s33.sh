#!/bin/bash
# frame streams with the message passing on high-priority partner streams
cd "$(dirname "$0")/../.."
for extra in "" "--gnn-priority -1" "--gnn-priority -1" ""; do
timeout 200 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline --no-live-pmc --no-capture --no-roofline $extra 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('[%s] fps %.1f  ms %.3f' % ('$extra', d['value'], d['ms_per_step']))
"
done
