#!/bin/bash
# round 5, session 14: f16x2 pooling with the 64->128 layer in the two-part
# arithmetic as well: tests, same-box A/B (f16_pool = 1 / 2 / 0), full suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s14
timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_bf16x3.py -q -m gpu -s -p no:cacheprovider > $O.tests1.log 2>&1
echo "TESTS1 rc=$? $(tail -1 $O.tests1.log)"
grep -E "^(FAILED|ERROR)|pooling E|Error|error:" $O.tests1.log | head -20
for t in "" "--tune f16_pool=2" "--tune f16_pool=0"; do
  timeout 300 python bench.py --edge-arith f16x2 --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --steps 20 --warmup 5 $t > $O.bench.json 2> $O.bench.err
  echo "BENCH f16x2 [$t] rc=$? $(python - <<PY
import json
d=json.loads(open('$O.bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_frame_per_gpu'])
PY
)"
done
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O.tests2.log 2>&1
echo "TESTS2 rc=$? $(tail -1 $O.tests2.log)"
grep -E "^(FAILED|ERROR)" $O.tests2.log | head
