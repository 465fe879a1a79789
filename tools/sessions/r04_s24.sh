#!/bin/bash
# round 4, session 24: whole-frame streams 2 / 3 / 4 / 5 after the K-row work
cd "$GRAFT_REPO_ROOT"
for fs in 3 2 4 5 3; do
  python bench.py --frame-streams $fs --steps 20 --warmup 5 --repeats 3 --no-secondary --no-roofline --no-cpu-baseline --no-capture 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('frame streams $fs: %.1f frames/s  repeats %s' % (d['value'], [round(x,2) for x in c['repeat_ms_per_step']['all']]))"
done | tee gpurun_out/r04_s24_streams.txt
