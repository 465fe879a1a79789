#!/bin/bash
# Kernel A/B on ONE GPU box: the same frames through two builds of the library
# (PGNN_LIB), per-kernel averages from rocprofv3 --kernel-trace --stats.
# usage: tools/ab.sh [libA.so] [libB.so] [kernel_bench args...]
#   defaults: ab/libbase.so (a build of HEAD, see DESIGN.md "A/B") vs the tree
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
A=${1:-$ROOT/ab/libbase.so}
B=${2:-$ROOT/point-gnn_amd/libpointgnn_hip.so}
shift 2 2>/dev/null
ARGS=${*:-frame --reps 40}
OUT=$ROOT/gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for round in 1 2; do
  for tag in A B; do
    lib=$A; [[ $tag == B ]] && lib=$B
    (cd /tmp && PGNN_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats \
        -d $OUT/$tag$round -o run -- python $ROOT/tools/kernel_bench.py $ARGS \
        > $OUT/$tag$round.log 2>&1)
    db=$(find $OUT/$tag$round -name "*.db" | head -1)
    echo "== $tag$round ($lib)"
    python - "$db" <<'EOF'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, tot, avg, pct in db.execute(
        "select name, total_calls, total_duration, average, percentage "
        "from top_kernels order by total_duration desc limit 7"):
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("  %-60s calls %5d avg %9.2f us  %5.1f%%" % (short, calls, avg, pct))
EOF
    rm -rf $OUT/$tag$round
  done
done
