#!/bin/bash
# round 4, session 19: chip-wide balanced column-group partition (fp32 + bf16x3)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16x3.py tests/test_gpu_train.py -x -q -m gpu -k "weights_stationary or segmax_epilogue or full_size_logits or edge_stage or edge_rows or native_sparse" 2>&1 | tail -4
for bal in 1 0; do
  echo "== ws_balance=$bal"
  PGNN_TUNE_WS_BALANCE=$bal timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, runpy
sys.argv = ["x"]
import pointgnn_amd
from pointgnn_amd import _lib
_lib.load(); _lib.set_tunable("ws_balance", $bal)
runpy.run_path("tools/bf16x3_bench.py", run_name="__main__")
PY
done
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r04_s19_bench.json 2> gpurun_out/r04_s19_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s19_bench.json').read().strip().splitlines()[-1])
c=d['config']
print(d['value'], d['ms_per_step'], c['repeat_ms_per_step']['all'])
print('edge', d['roofline_mfma']['avg_launch_us'], d['roofline_mfma']['frac'], 'pool', d['roofline_pool']['avg_launch_us'])
print('car', c['secondary']['frames_per_sec'], 'ped', c['secondary_ped']['frames_per_sec'], 'train', c['secondary_train']['ms_per_step'])
print('bf16x3', c['secondary_bf16x3'])
PY
