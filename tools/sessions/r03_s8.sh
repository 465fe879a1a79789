#!/bin/bash
# bench A/B over extra bench arguments (one quoted string per run; a leading
# ENV=VALUE word is exported for that run)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s8
mkdir -p $OUT
i=0
for extra in "$@"; do
  i=$((i+1))
  envs=""
  args="$extra"
  while [[ "$args" =~ ^([A-Z_]+=[^ ]+)\ ?(.*)$ ]]; do envs="$envs ${BASH_REMATCH[1]}"; args="${BASH_REMATCH[2]}"; done
  env $envs timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-secondary --no-roofline --steps 64 $args 2>$OUT/err_$i.log > $OUT/bench_$i.json
  python - <<PY
import json
try:
    b=json.load(open("$OUT/bench_$i.json")); c=b["config"]
    print("[$extra]: frames/s %.1f  ms/frame %.3f  overflow %s" % (b["value"], b["ms_per_step"], c.get("capacity_overflow_rebuilds")))
except Exception as e:
    print("[$extra] failed", e); print(open("$OUT/err_$i.log").read()[-1500:])
PY
done
