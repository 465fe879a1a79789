#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  echo "== $*"
  timeout 300 env $1 python bench.py --no-cpu-baseline --no-secondary --steps 64 $2 2>gpurun_out/s25_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f)' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))" \
    || tail -5 gpurun_out/s25_err.log
}
run A=1 ""
run A=1 "--lookahead 1"
run A=1 "--lookahead 2"
run A=1 "--lookahead 3"
run A=1 "--lookahead 2 --tune ws_reserve=8"
run A=1 "--tune graph_debug=1"
run A=1 ""
