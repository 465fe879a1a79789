#!/bin/bash
# round 5, session 21: where a ped_cyl frame's time goes (one frame, un-pipelined)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
ROOT=$PWD
O=$ROOT/gpurun_out/r05
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ped -o run -- python $ROOT/bench.py --config ped_cyl_auto_T3 --steps 8 --warmup 2 --frames 1 --no-cpu-baseline --no-secondary --no-live-pmc --no-pipeline --no-roofline > $O/prof_ped.log 2>&1)
db=$(find $O/prof_ped -name "*.db" | head -1)
python tools/prof_summary.py "$db" $O/ped_seed0_kernel_stats > /dev/null 2>&1
rm -rf $O/prof_ped
head -24 $O/ped_seed0_kernel_stats.md | cut -c1-120
