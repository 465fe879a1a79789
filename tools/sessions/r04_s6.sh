#!/bin/bash
# round 4, session 6: K-row kernels with every phase's HBM loads batched
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python tools/krow_bench.py > gpurun_out/r04_s6_krow.txt 2>&1
cat gpurun_out/r04_s6_krow.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "update_pre_edge or mlp2 or fused_vertex or vertex_pre_edge or mlp_forward or full_size_logits or predict_end_to_end" 2>&1 | tail -5 > gpurun_out/r04_s6_tests.log
tail -5 gpurun_out/r04_s6_tests.log
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r04_s6_bench.json 2> gpurun_out/r04_s6_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s6_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'], c['repeat_ms_per_step']['all'], c['host_enqueue_ms_per_frame'])
print(d['roofline_mfma']['avg_launch_us'], d['roofline_pool']['avg_launch_us'])
print(c['secondary']['frames_per_sec'], c['secondary_ped']['frames_per_sec'], c['secondary_train']['ms_per_step'])
PY
