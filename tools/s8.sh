#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s8
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for tag in kdon kdoff; do
  dbg=0; [[ $tag == kdoff ]] && dbg=1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/$tag -o run -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 40 --warmup 8 --lookahead 0 --tune graph_debug=$dbg > $OUT/$tag.log 2>&1)
  tail -1 $OUT/$tag.log | cut -c1-200
  db=$(find $OUT/$tag -name "*.db" | head -1)
  python tools/timeline_dump.py "$db" $OUT/$tag.csv
  rm -rf $OUT/$tag
done
ls -la $OUT
