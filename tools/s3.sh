#!/bin/bash
# session 3: parity of the new scheduling + aux-stream kd build, A/B of chunking
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
for c in 0 5 10 0 5 10; do
  echo "== chunks_per_wg $c"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 96 --tune mlp_chunks_per_wg=$c > gpurun_out/s3/bench_c$c.json 2> gpurun_out/s3/bench_c$c.err
  python - gpurun_out/s3/bench_c$c.json <<'PY'
import json,sys
b=json.load(open(sys.argv[1]))
print("frames/s %.1f  phase %s  edge_us %.1f pool_us %.1f" % (b["value"], b["config"]["phase_ms_frame_seed0"], b["roofline_mfma"]["avg_launch_us"], b["roofline_pool"]["avg_launch_us"]))
PY
done
