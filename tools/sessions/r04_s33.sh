#!/bin/bash
# round 4, session 33: box-to-box check of the latency-bound kernels (the final
# session's box ran them ~28 % slower than the mid-round boxes, untouched
# kernels like radix_scatter_kernel included)
cd "$GRAFT_REPO_ROOT"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -6
timeout 300 python tools/krow_bench.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r04_s33_krow.txt
for rep in 1 2; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('train: %.3f ms/step  %.1f frames/s  shape %s' % (d['ms_per_step'], d['value'], c['last_batch_shape']))"
done | tee gpurun_out/r04_s33_train.txt
