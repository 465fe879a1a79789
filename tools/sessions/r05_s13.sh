#!/bin/bash
# round 5, session 13: capacity-form fan-in cap + one-read training fetch;
# f16x2 pooling kernel (pool_ws_f16.h): tests, then the same-box A/B of whole
# frames with / without it
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s13
timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_bf16x3.py tests/test_gpu_e2e.py -q -m gpu -s -p no:cacheprovider > $O.tests1.log 2>&1
echo "TESTS1 rc=$? $(tail -1 $O.tests1.log)"
grep -E "^(FAILED|ERROR)|pooling E|Error|error:" $O.tests1.log | head -20
for t in "" "--tune f16_pool=0"; do
  timeout 300 python bench.py --edge-arith f16x2 --no-cpu-baseline --no-live-pmc --no-secondary --steps 20 --warmup 5 $t > $O.bench.json 2> $O.bench.err
  echo "BENCH f16x2 [$t] rc=$? $(python - <<PY
import json
d=json.loads(open('$O.bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_frame_per_gpu'], {k:(round(v.get('avg_launch_us',0),1), round(v.get('frac',0),3)) for k,v in d.items() if k.startswith('roofline') and isinstance(v,dict) and 'frac' in v})
PY
)"
done
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $O.tests2.log 2>&1
echo "TESTS2 rc=$? $(tail -1 $O.tests2.log)"
grep -E "^(FAILED|ERROR)" $O.tests2.log | head
