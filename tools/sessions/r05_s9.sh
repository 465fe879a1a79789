#!/bin/bash
# round 5, session 9: edge_arith-parametrised parity tests + the bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo skip tests
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r05_s9_bench.json 2> gpurun_out/r05_s9_bench.err; tail -3 gpurun_out/r05_s9_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_s9_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'])
print('edge', d['roofline_mfma']['avg_launch_us'], d['roofline_mfma']['frac'], 'pool', d['roofline_pool']['avg_launch_us'])
print('car', c['secondary']['frames_per_sec'], 'ped', c['secondary_ped']['frames_per_sec'], 'train', c['secondary_train']['ms_per_step'])
print('ped bf16x3', c['secondary_ped'].get('bf16x3'))
print('bf16x3', c['secondary_bf16x3'])
PY
