#!/bin/bash
# round 4, session 21: the default line after reordering the secondaries
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04_s21_bench.json 2> gpurun_out/r04_s21_bench.err
grep real gpurun_out/r04_s21_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s21_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'], c['repeat_ms_per_step']['all'])
print('edge', d['roofline_mfma']['avg_launch_us'], d['roofline_mfma']['frac'], 'pool', d['roofline_pool']['avg_launch_us'], d['roofline_pool']['frac'], 'scatter', d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('car', c['secondary']['frames_per_sec'], 'ped', c['secondary_ped']['frames_per_sec'], 'train', c['secondary_train']['ms_per_step'])
b=c['secondary_bf16x3']; print('bf16x3', b['frames_per_sec'], b['vs_f32_headline'], b['max_abs_dlogit_vs_f32_path_frame_seed0'])
PY
