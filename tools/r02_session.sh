#!/bin/bash
# Round-2 GPU session (via gpurun): whole GPU suite, smoke, bench lines and
# rocprofv3 kernel stats.  Everything lands in gpurun_out/r02/.
# usage: tools/r02_session.sh [tests] [bench] [prof] [pmc]   (default: all but pmc)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
MODES=${*:-tests bench prof}
S=$OUT/summary.txt
: > $S
echo "nproc=$(nproc) mem=$(free -g | awk '/Mem/{print $2}')G $(lscpu | grep 'Model name' | sed 's/  */ /g')" >> $S
for m in $MODES; do
case $m in
tests)
  timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1
  echo "TESTS rc=$? $(tail -1 $OUT/tests.log)" >> $S
  grep -E "^(FAILED|ERROR)" $OUT/tests.log | head -30 >> $S
  grep -E "reference TF graph|max\|dlogit\||worst" $OUT/tests.log | cut -c1-400 >> $S
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
  echo "SMOKE rc=$? $(grep '\[smoke\]' $OUT/smoke.log | tail -1)" >> $S
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
  echo "BENCH rc=$? $(head -c 6000 $OUT/bench.json)" >> $S; tail -3 $OUT/bench.err >> $S
  timeout 600 python bench.py --config ped_cyl_auto_T3 --no-cpu-baseline --steps 24 > $OUT/bench_ped.json 2> $OUT/bench_ped.err
  echo "BENCH_PED rc=$? $(head -c 2500 $OUT/bench_ped.json)" >> $S
  timeout 600 python bench.py --train --no-cpu-baseline --steps 24 > $OUT/bench_train.json 2> $OUT/bench_train.err
  echo "BENCH_TRAIN rc=$? $(head -c 2500 $OUT/bench_train.json)" >> $S
  ;;
prof)
  for what in bench infer; do
    case $what in
      bench) CMD="python $ROOT/bench.py --no-cpu-baseline --no-secondary";;
      infer) CMD="python $ROOT/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline";;
    esac
    rm -rf $OUT/prof_$what
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$what -o run -- $CMD > $OUT/prof_$what.log 2>&1)
    echo "PROF $what rc=$?" >> $S
    db=$(find $OUT/prof_$what -name "*.db" | head -1)
    python tools/prof_summary.py "$db" $OUT/${what}_kernel_stats > /dev/null 2>> $S
    rm -rf $OUT/prof_$what
    head -14 $OUT/${what}_kernel_stats.md | cut -c1-200 >> $S
  done
  ;;
pmc)
  bash tools/pmc_scatter.sh >> $S 2>&1
  ;;
esac
done
cat $S
