#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
for tag in kd nokd; do
  extra=""; [[ $tag == nokd ]] && extra="--tune graph_debug=1"
  rm -rf gpurun_out/tl_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl_$tag -o run -- python $ROOT/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline $extra > $ROOT/gpurun_out/tl_$tag.log 2>&1)
  db=$(find gpurun_out/tl_$tag -name "*.db" | head -1)
  python tools/timeline_dump.py "$db" gpurun_out/tl_$tag.csv 2>/dev/null
  rm -rf gpurun_out/tl_$tag
  tail -c 300 gpurun_out/tl_$tag.log | head -c 300; echo
  wc -l gpurun_out/tl_$tag.csv
done
