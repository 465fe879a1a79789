#!/bin/bash
# round 4, session 34: is the 3.15 ms step still hidden behind... rather, is the
# graph build of the next batch still hidden behind the (now shorter) step?
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for m in stream prebuilt; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 --train-loader $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('loader $m: %.3f ms/step  %.1f frames/s  shape %s' % (d['ms_per_step'], d['value'], c['last_batch_shape']))"
done
done | tee gpurun_out/r04_s34_train.txt
