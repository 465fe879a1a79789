#!/bin/bash
# Runs on the GPU box (via gpurun): every GPU test function in its own process
# (a kernel fault then costs one test, not the session), then smoke, bench and
# an optional rocprofv3 kernel trace.  Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [tests|bench|prof|all] (default all)
set -u
MODE=${1:-all}
OUT=gpurun_out
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
SUMMARY=$OUT/summary.txt
: > $SUMMARY
rocm-smi --showproductname 2>/dev/null | head -8 >> $SUMMARY
echo "nproc=$(nproc) mem=$(free -g | awk '/Mem/{print $2}')G" >> $SUMMARY
lscpu | grep -E "Model name|^CPU\(s\)" >> $SUMMARY

if [[ $MODE == all || $MODE == tests ]]; then
  python -c "import torch; print('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" >> $SUMMARY 2>&1
  TESTS=$(python -m pytest tests/test_gpu_parity.py --collect-only -q -m gpu 2>/dev/null | grep '::' | sed 's/\[.*//' | sort -u)
  for t in $TESTS; do
    name=$(echo $t | sed 's/.*:://')
    timeout 600 python -m pytest "$t" -q -m gpu -x -s > $OUT/test_$name.log 2>&1
    rc=$?
    echo "TEST $name rc=$rc $(tail -1 $OUT/test_$name.log)" >> $SUMMARY
    if [[ $rc != 0 ]]; then
      grep -E "^E |Error|error|fault|assert" $OUT/test_$name.log | head -12 | sed 's/^/    /' >> $SUMMARY
    fi
  done
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
  echo "SMOKE rc=$? $(grep '\[smoke\]' $OUT/smoke.log | tail -1)" >> $SUMMARY
fi

if [[ $MODE == all || $MODE == bench ]]; then
  timeout 900 python bench.py --steps 32 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err
  echo "BENCH rc=$? $(cat $OUT/bench.json | head -c 3000)" >> $SUMMARY
  tail -3 $OUT/bench.err >> $SUMMARY
fi

if [[ $MODE == all || $MODE == prof ]]; then
  rm -rf $OUT/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 16 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof.log 2>&1)
  echo "PROF rc=$?" >> $SUMMARY
  f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1)
  if [[ -n "$f" ]]; then
    cp "$f" $OUT/kernel_stats.csv
    head -25 "$f" | cut -c1-220 >> $SUMMARY
  fi
  # keep the merged-back directory small
  find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
fi
cat $SUMMARY
