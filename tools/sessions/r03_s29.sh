#!/bin/bash
# overlapped graph build (level-0 grid + kd-tree replica beside the voxel hash,
# level-0 queries beside the level-1 graph): parity tests, then the latency /
# build-alone entries of the bench line, eager and inside one hipGraph
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r03
for extra in "" "--config ped_cyl_auto_T3"; do
timeout 200 python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-live-pmc $extra 2>gpurun_out/r03/s29.err | python -c "
import json,sys
d=json.load(sys.stdin); c=d['config']
print('fps %.1f' % d['value'])
for k in c:
    if k.startswith('latency_ms') or k.startswith('graph_build_ms') or k.startswith('phase_ms'):
        print(k, json.dumps(c[k]))
" || tail -5 gpurun_out/r03/s29.err
done
