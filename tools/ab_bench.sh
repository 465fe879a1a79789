#!/bin/bash
# Same-box A/B of bench.py configurations: each argument is one run, written
# as "ENV=VALUE[,ENV=VALUE] -- bench args", e.g.
#   tools/ab_bench.sh "A=1 --" "A=1 -- --tune mlp_debug=2048" \
#                     "PGNN_LIB=$PWD/ab/libsched2.so --"
# (PGNN_LIB selects a variant build made by tools/build_variant.py).  Prints
# frames/s and the standalone edge / pooling kernel times of every run; run
# the baseline first AND last to see the box-to-box drift.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  envs=${spec%%--*}; args=${spec#*--}
  echo "== $spec"
  timeout 300 env ${envs//,/ } python bench.py --no-cpu-baseline --no-secondary --steps 64 $args 2>gpurun_out/ab_bench_err.log \
    | python -c "import json,sys; b=json.load(sys.stdin); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f) gen_graph_ms %.3f' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac'], b['config']['phase_ms_frame_seed0']['gen graph']))" \
    || tail -5 gpurun_out/ab_bench_err.log
done
