#!/bin/bash
# per-vertex kernel durations (sequential frame, rocprof) and bench fps per library variant
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s19
mkdir -p $OUT
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset PGNN_LIB; else export PGNN_LIB=$ROOT/ab/lib$lib.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python $ROOT/bench.py --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --no-pipeline --frames 1 --steps 12 --warmup 3 > $OUT/run.log 2>&1)
  db=$(find $OUT/prof -name "*.db" | head -1)
  python tools/prof_summary.py "$db" $OUT/stats_$lib > /dev/null
  rm -rf $OUT/prof
  echo "== $lib"; grep "rows_mlp\|vertex_pre_edge\|edge_ws\|pool_ws" $OUT/stats_$lib.md | cut -c1-100
  timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --steps 64 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('   frames/s %.1f' % d['value'])"
done
