#!/bin/bash
# round 5, session 19: winners' count pass of the sparse adjoint, items in
# flight per thread (2 / 4 / 8): same-box A/B of the training step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s19
for rep in 1 2; do
for v in "" cw2 cw8; do
  L=${v:+ab/lib$v.so}
  PGNN_LIB=$L timeout 300 python bench.py --train --steps 24 --warmup 8 --frames 4 --no-live-pmc > $O.b.json 2> $O.b.err
  echo "TRAIN ${v:-cw4} rc=$? $(python - <<PY
import json
d=json.loads(open('$O.b.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
PY
)"
done
done
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
