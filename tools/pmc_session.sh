#!/bin/bash
# PMC passes (each counter set in its own run, kernel-trace only) for the two
# roofline kernels.  Results: gpurun_out/pmc_*.db
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # name, counters, what
  rm -rf $OUT/pmc_$1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $OUT/pmc_$1 -o pmc -- python $OLDPWD/tools/kernel_bench.py $3 --reps 5 > $OUT/pmc_$1.log 2>&1)
  echo "PMC $1 rc=$?"
  ls $OUT/pmc_$1 2>/dev/null
}
run scatter_fetch "FETCH_SIZE" scatter
run scatter_write "WRITE_SIZE" scatter
run edge_fetch "FETCH_SIZE" edge
run edge_write "WRITE_SIZE" edge
run edge_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" edge
run edge_sq2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" edge
