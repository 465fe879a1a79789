#!/bin/bash
# round 4, session 35: the slice reduce of pool_narrow_bwd with 16 slice groups
# per workgroup and four loads in flight per thread (was 3 x ~50 us per step)
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r04_s35_tests.txt
for rep in 1 2; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('train: %.3f ms/step  %.1f frames/s  shape %s' % (d['ms_per_step'], d['value'], c['last_batch_shape']))"
done | tee gpurun_out/r04_s35_train.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04_s35_prof -o run -- python $GRAFT_REPO_ROOT/bench.py --train --steps 8 --warmup 4 --frames 4 > $OUT/r04_s35_prof.log 2>&1)
db=$(find $OUT/r04_s35_prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" $OUT/r04_s35_train_kernel_stats > /dev/null
rm -rf $OUT/r04_s35_prof
grep -i "reduce\|radix_scatter\|rows_mlp" $OUT/r04_s35_train_kernel_stats.md | cut -c1-120
