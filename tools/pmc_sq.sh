#!/bin/bash
# SQ-level PMC passes for the MFMA-bound kernels (edge: edge_ws_kernel, or
# fused_mlp_kernel<4, 2> under PGNN_TUNE="mlp_debug=2048"; pooling:
# fused_mlp_kernel<4, 1|3>) on whole frames of a preset.  One counter set
# per run, kernel-trace only (gpurun refuses --pmc with other trace domains; the
# TA_*/TCP_* sets abort rocprofv3 on this pool -- do not add them).
# usage: tools/pmc_sq.sh [preset] ; prints per-launch averages, writes
# gpurun_out/pmc_sq_<preset>.txt
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
PRESET=${1:-car_600k}
OUT=$ROOT/gpurun_out/pmc_sq_work
RES=$ROOT/gpurun_out/pmc_sq_$PRESET.txt
rm -rf $OUT; mkdir -p $OUT
: > $RES
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LEVEL_WAVES SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  [[ -n "${PMC_SETS:-}" && $i -gt $PMC_SETS ]] && break
  # PMC_CMD overrides the profiled command (e.g. the training bench)
  CMD=${PMC_CMD:-"python $ROOT/tools/kernel_bench.py frame --reps 6 --preset $PRESET ${PGNN_TUNE:+--tune $PGNN_TUNE}"}
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- \
      $CMD > $OUT/p$i.log 2>&1)
  echo "set $i rc=$?: $set" >> $RES
  db=$(find $OUT/p$i -name "*.db" | head -1)
  python - "$db" >> $RES <<'EOF2'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for pat, tag in (("edge_ws_kernel", "edgews"), ("pool_ws_kernel", "poolws"),
                 ("fused_mlp_kernel<4, 2>", "edge"),
                 ("fused_mlp_kernel<4, 1>", "pool"), ("fused_mlp_kernel<4, 3>", "pool"),
                 ("weight_grad_kernel<5>", "wgrad5"), ("weight_grad_kernel<2>", "wgrad2"),
                 ("segmax_route_scatter_kernel", "route"),
                 ("fused_mlp_kernel<4, 0>", "rowsE")):
    rows = db.execute("select counter_name, count(*), avg(value) from counters_collection "
                      "where kernel_name like ? group by counter_name", ("%" + pat + "%",))
    n = db.execute("select count(distinct dispatch_id) from counters_collection where "
                   "kernel_name like ?", ("%" + pat + "%",)).fetchone()[0]
    for c, cnt, avg in rows:
        print("  %-5s %-34s dispatches %3d  avg/launch %.4e" % (tag, c, n, avg))
EOF2
  rm -rf $OUT/p$i
done
cat $RES
