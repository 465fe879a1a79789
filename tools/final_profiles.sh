#!/bin/bash
# Round-end evidence run on the GPU box: bench JSON (inference + training),
# rocprofv3 kernel stats of both and of the detection stage -> gpurun_out/final/
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err
for what in bench infer train detect; do
  case $what in
    bench) CMD="python $ROOT/bench.py --no-cpu-baseline";;
    infer) CMD="python $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-pipeline";;
    train) CMD="python $ROOT/bench.py --train --steps 6 --warmup 2 --no-cpu-baseline";;
    detect) CMD="python $ROOT/tools/kernel_bench.py detect";;
  esac
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$what -o run -- $CMD > $OUT/prof_$what.log 2>&1)
  db=$(find $OUT/prof_$what -name "*.db" | head -1)
  python tools/prof_summary.py "$db" $OUT/${what}_kernel_stats > /dev/null
  rm -rf $OUT/prof_$what
done
tail -c 600 $OUT/bench.json; echo; tail -c 400 $OUT/bench_train.json; echo
head -12 $OUT/infer_kernel_stats.md
