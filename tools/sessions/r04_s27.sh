#!/bin/bash
# round 4, session 27: training step after the launch diet (ReluGrad in the dX
# epilogue, dP|dQ one fill, no tail fill, one dst column per level, one box
# slice kernel) -- train tests, then the step timed
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_tfgraph.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_s27_tests.txt
for rep in 1 2; do
for m in "--train-loader stream" "--train-loader stream --train-sync-loss"; do
  timeout 300 python bench.py --train --steps 24 --warmup 8 $m 2>gpurun_out/r04_s27.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$m: %.3f ms/step  %.1f frames/s  shape %s loss %s' % (d['ms_per_step'], d['value'], c['last_batch_shape'], c['last_loss']))"
done
done | tee gpurun_out/r04_s27_train.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r04_s27_prof -o run -- python $GRAFT_REPO_ROOT/bench.py --train --steps 8 --warmup 4 --frames 4 > $OUT/r04_s27_prof.log 2>&1)
db=$(find $OUT/r04_s27_prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" $OUT/r04_s27_train_kernel_stats > /dev/null
python tools/trace_dump.py "$db" --last-ms 11 --out $OUT/r04_s27_train_trace.txt
rm -rf $OUT/r04_s27_prof
head -30 $OUT/r04_s27_train_kernel_stats.md | cut -c1-120
