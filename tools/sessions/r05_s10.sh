#!/bin/bash
# round 5, session 10: the routed scatter-max adjoint with batched winners +
# row prefetch: gradient tests, step time and per-kernel averages A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -4
export TMPDIR=/tmp
for tag in base new wb4; do
  lib=$PWD/point-gnn_amd/libpointgnn_hip.so
  [[ $tag == base ]] && lib=$PWD/ab/libbase_train.so
  [[ $tag == wb4 ]] && lib=$PWD/ab/libwb4.so
  echo "== $tag"
  PGNN_LIB=$lib python bench.py --train --steps 24 --warmup 8 --no-live-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'])"
  (cd /tmp && PGNN_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --train --steps 12 --warmup 4 --no-live-pmc --train-loader prebuilt > /dev/null 2>&1)
  db=$(find gpurun_out/tr_$tag -name "*.db" | head -1)
  python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, tot, avg, pct in db.execute(
        "select name, total_calls, total_duration, average, percentage "
        "from top_kernels order by total_duration desc limit 8"):
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("  %-60s calls %5d avg %9.2f us  %5.1f%%" % (short, calls, avg / 1e3 if avg > 1e4 else avg, pct))
PY
  rm -rf gpurun_out/tr_$tag
done
