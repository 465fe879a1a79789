#!/bin/bash
# SQ-only PMC passes for the fused edge kernel (each counter set in its own run,
# kernel-trace only -- gpurun refuses --pmc together with other trace domains).
# Prints per-launch averages of every counter for fused_mlp_kernel<4, 2>.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_sq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  (cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- \
      python $ROOT/tools/kernel_bench.py frame --reps 6 > $OUT/p$i.log 2>&1)
  db=$(find $OUT/p$i -name "*.db" | head -1)
  python - "$db" <<'EOF'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
views = [r[0] for r in db.execute("select name from sqlite_master where type='view'")]
v = "counters_collection" if "counters_collection" in views else None
if v is None:
    print("no counters_collection view:", views); sys.exit(0)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % v)]
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = db.execute("select counter_name, count(*), avg(value), sum(value) from %s "
                  "where %s like '%%fused_mlp_kernel<4, 2>%%' group by counter_name" % (v, name_col))
for c, n, avg, tot in rows:
    print("  %-32s rows %6d  avg/row %.4e  total %.4e" % (c, n, avg, tot))
print("  dispatches:", db.execute("select count(distinct dispatch_id) from %s where %s like "
      "'%%fused_mlp_kernel<4, 2>%%'" % (v, name_col)).fetchone()[0])
EOF
  rm -rf $OUT/p$i
done
