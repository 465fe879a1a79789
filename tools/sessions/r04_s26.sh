#!/bin/bash
# round 4, session 26: training step with its tail queued before the host read
# (L1, SGD, repack, then one pinned copy) and the losses read one step late
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_s26_tests.txt
for rep in 1 2; do
for m in "--train-loader stream" "--train-loader stream --train-sync-loss" "--train-loader prebuilt" "--train-loader prebuilt --train-sync-loss"; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 $m 2>gpurun_out/r04_s26.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$m: %.3f ms/step  %.1f frames/s  shape %s loss %s' % (d['ms_per_step'], d['value'], c['last_batch_shape'], c['last_loss']))"
done
done | tee gpurun_out/r04_s26_train.txt
tail -3 gpurun_out/r04_s26.err
