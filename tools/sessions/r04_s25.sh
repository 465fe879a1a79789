#!/bin/bash
# round 4, session 25: the training step with the next batch built (a) by the
# stepping thread on a second stream, (b) by a loader thread two batches ahead,
# (c) not at all (prebuilt batches: the step without its data side)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for m in stream thread prebuilt; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 --train-loader $m 2>gpurun_out/r04_s25_$m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('loader $m: %.3f ms/step  %.1f frames/s  shape %s loss %s' % (d['ms_per_step'], d['value'], c['last_batch_shape'], c['last_loss']))"
done
done | tee gpurun_out/r04_s25_loader.txt
tail -3 gpurun_out/r04_s25_thread.err
