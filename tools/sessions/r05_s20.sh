#!/bin/bash
# round 5, session 20: f16x2 edge kernel with the range guard in its prologue
# (one pass over P / Q) instead of a running maximum in the MFMA loop: kernel
# and whole-frame A/B against the previous commit (ab/libbase.so), tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s20
for rep in 1 2; do
for v in "" base; do
  L=${v:+ab/lib$v.so}
  echo "== ${v:-new}"
  PGNN_LIB=$L timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "fp16 x2"
done
done
for v in "" base; do
  L=${v:+ab/lib$v.so}
  PGNN_LIB=$L timeout 300 python bench.py --edge-arith f16x2 --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --steps 20 --warmup 5 > $O.bench.json 2> $O.bench.err
  echo "BENCH f16x2 [${v:-new}] rc=$? $(python - <<PY
import json
d=json.loads(open('$O.bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_frame_per_gpu'])
PY
)"
done
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_deferred.py tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
