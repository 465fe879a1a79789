#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6
run() {
  echo "== $*"
  timeout 300 env $1 python bench.py --no-cpu-baseline --no-secondary --steps 64 $2 2>gpurun_out/s23_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f) phases %s' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac'], b['config']['phase_ms_frame_seed0']))" \
    || tail -5 gpurun_out/s23_err.log
}
run A=1 ""
run A=1 ""
rm -rf gpurun_out/s23_prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/s23_prof -o run -- python $OLDPWD/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline > $OLDPWD/gpurun_out/s23_prof.log 2>&1)
db=$(find gpurun_out/s23_prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" gpurun_out/s23_infer_kernel_stats > /dev/null; head -22 gpurun_out/s23_infer_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/s23_prof
