#!/bin/bash
# round 5, session 16: pool_narrow_bwd with the next tile's rows requested
# under this tile's work: training tests, same-box A/B of the step
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s16
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_multilevel.py -q -m gpu -p no:cacheprovider > $O.tests.log 2>&1
echo "TESTS rc=$? $(tail -1 $O.tests.log)"
grep -E "^(FAILED|ERROR)" $O.tests.log | head
for rep in 1 2; do
for v in "" pn0; do
  L=${v:+ab/lib$v.so}
  PGNN_LIB=$L timeout 300 python bench.py --train --steps 24 --warmup 8 --frames 4 --no-live-pmc > $O.b.json 2> $O.b.err
  echo "TRAIN ${v:-new} rc=$? $(python - <<PY
import json
d=json.loads(open('$O.b.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'])
PY
)"
done
done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/s16prof -o run -- python $GRAFT_REPO_ROOT/bench.py --train --steps 8 --warmup 4 --frames 4 --no-live-pmc > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r05/s16prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" gpurun_out/r05/s16_train_kernel_stats > /dev/null 2>&1
grep -E "pool_narrow_bwd|pool_ws_kernel|edge_ws" gpurun_out/r05/s16_train_kernel_stats.md
rm -rf gpurun_out/r05/s16prof
