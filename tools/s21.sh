#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
  -k "weights_stationary or full_size_logits" 2>&1 | tail -4
run() {
  echo "== $*"
  timeout 300 env $1 python bench.py --no-cpu-baseline --no-secondary --steps 48 $2 2>gpurun_out/s21_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f)' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))" \
    || tail -5 gpurun_out/s21_err.log
}
run A=1 ""
run PGNN_LIB=$PWD/ab/libnogw.so ""
run A=1 ""
run PGNN_LIB=$PWD/ab/libnogw.so ""
run A=1 ""
