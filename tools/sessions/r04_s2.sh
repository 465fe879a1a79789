#!/bin/bash
# round 4, session 2: weights->LDS prologue with all requests in flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "weights_stationary or segmax_epilogue or full_size_logits" 2>&1 | tail -5 > gpurun_out/r04_s2_tests.log
python tools/ws_timeline.py --cost-model > gpurun_out/r04_s2_timeline.txt 2>&1
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r04_s2_bench.json 2> gpurun_out/r04_s2_bench.err
tail -3 gpurun_out/r04_s2_tests.log; grep -E "kernel|wave end|SIMD" gpurun_out/r04_s2_timeline.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s2_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'], c['repeat_ms_per_step'], c['host_enqueue_ms_per_frame'], c['cpu_affinity'])
print(d['roofline_mfma']['avg_launch_us'], d['roofline_pool']['avg_launch_us'])
print(c['secondary_ped']['frames_per_sec'], c['secondary_train']['ms_per_step'])
PY
