#!/bin/bash
# Round-6 evidence session (via gpurun) -> gpurun_out/r06/ (copy the summaries
# into profiles/r06_*).  usage: tools/r06_final.sh [tests] [bench] [prof] [pmc]
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
MODES=${*:-tests bench prof pmc}
S=$OUT/summary.txt
: > $S
echo "nproc=$(nproc) mem=$(free -g | awk '/Mem/{print $2}')G $(lscpu | grep 'Model name' | sed 's/  */ /g')" >> $S
for m in $MODES; do
case $m in
tests)
  timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1
  echo "TESTS rc=$? $(tail -1 $OUT/tests.log)" >> $S
  grep -E "^(FAILED|ERROR)" $OUT/tests.log | head -30 >> $S
  grep -E "reference TF graph|max\|dlogit\||worst|max error vs float64" $OUT/tests.log | cut -c1-300 >> $S
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
  echo "SMOKE rc=$? $(grep '\[smoke\]' $OUT/smoke.log | tail -1)" >> $S
  ;;
bench)
  ( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
  echo "BENCH rc=$? $(grep real $OUT/bench.err) $(head -c 1200 $OUT/bench.json)" >> $S
  timeout 600 python bench.py --config ped_cyl_auto_T3 --no-cpu-baseline --no-live-pmc --steps 24 > $OUT/bench_ped.json 2> $OUT/bench_ped.err
  echo "BENCH_PED rc=$? $(head -c 600 $OUT/bench_ped.json)" >> $S
  timeout 600 python bench.py --train --steps 24 --warmup 8 --frames 4 > $OUT/bench_train.json 2> $OUT/bench_train.err
  echo "BENCH_TRAIN rc=$? $(head -c 600 $OUT/bench_train.json)" >> $S
  timeout 600 python bench.py --train --steps 24 --warmup 8 > $OUT/bench_train8.json 2> $OUT/bench_train8.err
  echo "BENCH_TRAIN(8-frame pool) rc=$? $(head -c 400 $OUT/bench_train8.json)" >> $S
  timeout 600 python bench.py --train --steps 24 --warmup 8 --frames 4 --no-live-pmc --train-loader prebuilt > $OUT/bench_train_prebuilt.json 2> $OUT/bench_train_prebuilt.err
  echo "BENCH_TRAIN(prebuilt batches: the step without its data side) rc=$? $(head -c 300 $OUT/bench_train_prebuilt.json)" >> $S
  timeout 600 python bench.py --e2e > $OUT/bench_e2e.json 2> $OUT/bench_e2e.err
  echo "BENCH_E2E rc=$? $(head -c 900 $OUT/bench_e2e.json)" >> $S
  ;;
prof)
  for what in bench infer infer_seed0 ped_seed0 train e2e; do
    case $what in
      bench) CMD="python $ROOT/bench.py --no-cpu-baseline --no-secondary --no-live-pmc";;
      infer) CMD="python $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-secondary --no-live-pmc --no-pipeline";;
      infer_seed0) CMD="python $ROOT/bench.py --steps 8 --warmup 2 --frames 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-pipeline --no-roofline";;
      ped_seed0) CMD="python $ROOT/bench.py --config ped_cyl_auto_T3 --steps 6 --warmup 2 --frames 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-pipeline --no-roofline";;
      e2e) CMD="python $ROOT/bench.py --e2e";;
      train) CMD="python $ROOT/bench.py --train --steps 8 --warmup 4 --frames 4 --no-live-pmc";;
    esac
    rm -rf $OUT/prof_$what
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$what -o run -- $CMD > $OUT/prof_$what.log 2>&1)
    echo "PROF $what rc=$?" >> $S
    db=$(find $OUT/prof_$what -name "*.db" | head -1)
    python tools/prof_summary.py "$db" $OUT/${what}_kernel_stats > /dev/null 2>> $S
    if [ $what = train ]; then
      python tools/train_step_trace.py "$db" --out $OUT/train_step_trace.txt > $OUT/train_step_summary.txt
    fi
    rm -rf $OUT/prof_$what
    head -12 $OUT/${what}_kernel_stats.md | cut -c1-160 >> $S
  done
  ;;
pmc)
  PMC_SETS=1 bash tools/pmc_sq.sh car_600k > $OUT/pmc_sq_infer.log 2>&1
  cp gpurun_out/pmc_sq_car_600k.txt $OUT/pmc_sq_infer.txt 2>/dev/null
  PMC_SETS=1 PMC_CMD="python $ROOT/bench.py --train --steps 6 --warmup 3 --frames 4" bash tools/pmc_sq.sh train > $OUT/pmc_sq_train.log 2>&1
  cp gpurun_out/pmc_sq_train.txt $OUT/pmc_sq_train.txt 2>/dev/null
  cat $OUT/pmc_sq_train.txt >> $S
  ;;
esac
done
cat $S
