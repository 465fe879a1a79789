#!/bin/bash
# round 4, session 36: l1_norm with <= 128 atomics, coalesced 16-wide pool
# feature rows -- train tests, the step timed
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r04_s36_tests.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('train: %.3f ms/step  %.1f frames/s  shape %s' % (d['ms_per_step'], d['value'], c['last_batch_shape']))"
done | tee gpurun_out/r04_s36_train.txt
