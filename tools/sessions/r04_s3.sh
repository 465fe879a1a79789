#!/bin/bash
# round 4, session 3: fused per-vertex stages (bit-identity + rate) and the
# edge kernel's entry stamps
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "update_pre_edge or mlp2 or fused_vertex or vertex_pre_edge" 2>&1 | tail -15 > gpurun_out/r04_s3_tests.log
tail -5 gpurun_out/r04_s3_tests.log
python tools/ws_timeline.py > gpurun_out/r04_s3_timeline.txt 2>&1
grep -E "kernel|wave entry|wave end" gpurun_out/r04_s3_timeline.txt
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r04_s3_bench.json 2> gpurun_out/r04_s3_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s3_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'], c['repeat_ms_per_step']['all'], c['host_enqueue_ms_per_frame'])
print(d['roofline_mfma']['avg_launch_us'], d['roofline_pool']['avg_launch_us'])
print(c['secondary']['frames_per_sec'], c['secondary_ped']['frames_per_sec'], c['secondary_train']['ms_per_step'])
print(c['latency_ms_frame_seed0'], c['phase_ms_frame_seed0'])
PY
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r04_s3_prof" -o infer -- python "$GRAFT_REPO_ROOT/bench.py" --no-pipeline --frames 1 --steps 4 --warmup 1 --repeats 1 --no-roofline --no-cpu-baseline --no-secondary --no-capture > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; DB=$(find gpurun_out/r04_s3_prof -name "*.db" | head -1); echo "db: $DB"; python tools/prof_summary.py "$DB" gpurun_out/r04_s3_infer_seed0_kernel_stats > /dev/null 2>&1; head -16 gpurun_out/r04_s3_infer_seed0_kernel_stats.md; rm -rf gpurun_out/r04_s3_prof
