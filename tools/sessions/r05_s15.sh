#!/bin/bash
# round 5, session 15: f16x2 edge kernel, request distances of the raw rows
# (PGNN_F16_DP / _DQ blocks ahead; base 3 / 2): same-box kernel A/B; the tests
# touched by the f16x2 pooling stage
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s15
for v in "" f16_43 f16_54 f16_64 f16_65; do
  L=${v:+ab/lib$v.so}
  echo "== car_600k ${v:-base}"
  PGNN_LIB=$L timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "fp32-MFMA|fp16 x2"
done
for v in "" f16_54 f16_65; do
  L=${v:+ab/lib$v.so}
  echo "== ped_dense ${v:-base}"
  PGNN_LIB=$L timeout 200 python tools/bf16x3_bench.py --preset ped_dense --config ped_cyl_auto_T3 2>&1 | grep -E "fp32-MFMA|fp16 x2"
done
timeout 900 python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_tfgraph.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider > $O.tests.log 2>&1
echo "TESTS rc=$? $(tail -1 $O.tests.log)"
grep -E "^(FAILED|ERROR)" $O.tests.log | head
