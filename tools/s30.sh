#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  echo "== $*"
  timeout 300 env $1 python bench.py --no-cpu-baseline --no-secondary --steps 64 $2 2>gpurun_out/s30_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f pool_us %.1f gen_graph_ms %.3f' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_pool']['avg_launch_us'], b['config']['phase_ms_frame_seed0']['gen graph']))" \
    || tail -5 gpurun_out/s30_err.log
}
run A=1 ""
run A=1 "--lookahead 2 --graph-streams 2"
run A=1 "--lookahead 3 --graph-streams 2"
run A=1 "--lookahead 1 --graph-streams 2"
run A=1 "--lookahead 2 --graph-streams 2 --tune graph_debug=1"
run A=1 ""
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pipelined" 2>&1 | tail -3
