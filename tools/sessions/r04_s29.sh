#!/bin/bash
# round 4, session 29: training step with the K-row MLP chains in one launch
# each (forward: outputs tapped; backward: dX chain, ReluGrad in LDS), and the
# A/B of (a) H1 not materialised, (b) the balanced column-group partition
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_s29_tests.txt
for rep in 1 2; do
for m in "" "train_h1=0" "ws_balance=2" "train_h1=0,ws_balance=2"; do
  PGNN_TUNE="$m" timeout 300 python bench.py --train --steps 24 --warmup 8 2>gpurun_out/r04_s29.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('tune [$m]: %.3f ms/step  %.1f frames/s  shape %s loss %s' % (d['ms_per_step'], d['value'], c['last_batch_shape'], c['last_loss']))"
done
done | tee gpurun_out/r04_s29_train.txt
PGNN_TUNE="train_h1=0" timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/r04_s29_tests.txt
