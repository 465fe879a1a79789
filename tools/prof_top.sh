#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command, top kernels printed.
# usage: tools/prof_top.sh <n> <command...>
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
N=$1; shift
OUT=$ROOT/gpurun_out/prof_top
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o run -- "$@" > $OUT/cmd.log 2>&1)
tail -3 $OUT/cmd.log
db=$(find $OUT -name "*.db" | head -1)
python - "$db" "$N" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name, calls, tot, avg, pct in db.execute(
        "select name, total_calls, total_duration, average, percentage "
        "from top_kernels order by total_duration desc limit %d" % int(sys.argv[2])):
    short = name.replace("(anonymous namespace)::", "").split("(")[0][:64]
    print("  %-64s calls %6d avg %9.2f us  %5.1f%%" % (short, calls, avg, pct))
PY
