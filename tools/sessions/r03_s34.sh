#!/bin/bash
# more hardware queues (GPU_MAX_HW_QUEUES, default 4) for more frame streams?
cd "$(dirname "$0")/../.."
run() {
  env $1 timeout 100 python bench.py --steps 40 --warmup 8 --no-secondary --no-cpu-baseline --no-live-pmc --no-capture --no-roofline --frame-streams $2 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('[$1 streams $2] fps %.1f  ms %.3f' % (d['value'], d['ms_per_step']))
"
}
run A=1 3
run GPU_MAX_HW_QUEUES=8 4
run GPU_MAX_HW_QUEUES=8 5
run GPU_MAX_HW_QUEUES=8 3
