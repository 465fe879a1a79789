#!/bin/bash
# PMC passes for the standalone scatter-max kernel (final tuning): FETCH_SIZE and
# WRITE_SIZE each in its own run (TCC slot limit), kernel-trace only.
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc2_scatter_$c
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc2_scatter_$c -o pmc -- python $OLDPWD/tools/kernel_bench.py scatter --reps 5 --preset car_600k > $OUT/pmc2_scatter_$c.log 2>&1)
  echo "PMC scatter $c rc=$?"
done
