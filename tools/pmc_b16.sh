#!/bin/bash
# SQ counters of the 16-bit edge kernels (bf16x3, f16x2; the fp32 one beside them) on the
# bench frame's first GNN iteration (tools/bf16x3_bench.py): one counter set per
# rocprofv3 run, kernel-trace only.  usage: tools/pmc_b16.sh <result file> [sets]
cd "$(dirname "$0")/.."
ROOT=$PWD
RES=${1:-$ROOT/gpurun_out/pmc_sq_bf16x3.txt}
NSETS=${2:-5}
OUT=$ROOT/gpurun_out/pmc_b16_work; rm -rf $OUT; mkdir -p $OUT
: > $RES
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  [[ $i -gt $NSETS ]] && break
  (cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc -- python $ROOT/tools/bf16x3_bench.py > $OUT/p$i.log 2>&1)
  echo "set $i rc=$?: $set" >> $RES
  db=$(find $OUT/p$i -name "*.db" | head -1)
  python - "$db" >> $RES <<'EOF2'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for pat, tag in (("edge_ws_bf16x3_kernel", "b16x3"), ("edge_ws_f16x2_kernel", "f16x2"),
                 ("edge_ws_kernel", "f32")):
    rows = db.execute("select counter_name, count(*), avg(value) from counters_collection "
                      "where kernel_name like ? group by counter_name", ("%" + pat + "%",))
    for c, cnt, avg in rows:
        print("  %-6s %-34s n %3d  avg/launch %.4e" % (tag, c, cnt, avg))
EOF2
  rm -rf $OUT/p$i
done
cat $RES
