#!/bin/bash
# round 5, session 9: edge_arith-parametrised parity tests + the bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_tfgraph.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_deferred.py -x -q -m gpu -s -k "bf16x3 or f16x2 or predict_matches_reference_tf_graph or predict_real_weights or deferred_frame_is_bit_identical or frames_on_streams_equal_sequential" 2>&1 | grep -v "amdgpu.ids" | grep -i "f16x2\|passed\|failed\|error" | tail -40
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r05_s9_bench.json 2> gpurun_out/r05_s9_bench.err; tail -3 gpurun_out/r05_s9_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_s9_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'])
print('edge', d['roofline_mfma']['avg_launch_us'], d['roofline_mfma']['frac'], 'pool', d['roofline_pool']['avg_launch_us'])
print('car', c['secondary']['frames_per_sec'], 'ped', c['secondary_ped']['frames_per_sec'], 'train', c['secondary_train']['ms_per_step'])
print('ped bf16x3', c['secondary_ped'].get('bf16x3'))
print('bf16x3', {k: v for k, v in c['secondary_bf16x3'].items() if k in ('frames_per_sec', 'vs_f32_headline')}, c['secondary_bf16x3']['roofline']['avg_launch_us'], c['secondary_bf16x3']['roofline']['frac']); print('f16x2', {k: v for k, v in c['secondary_f16x2'].items() if 'seed' in k or k in ('frames_per_sec', 'vs_f32_headline')}, c['secondary_f16x2']['roofline']['avg_launch_us'], c['secondary_f16x2']['roofline']['frac']); print('ped f16x2', c['secondary_ped'].get('f16x2', {}).get('frames_per_sec'))
PY
