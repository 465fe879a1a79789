#!/bin/bash
# round 4, session 30: training step -- Wx-row weight gradients in a second
# deferred batch, dQ Wx^T as the first layer of the offset MLP's backward chain
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_s30_tests.txt
for rep in 1 2 3; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 2>gpurun_out/r04_s30.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%.3f ms/step  %.1f frames/s  shape %s loss %s' % (d['ms_per_step'], d['value'], c['last_batch_shape'], c['last_loss']))"
done | tee gpurun_out/r04_s30_train.txt
