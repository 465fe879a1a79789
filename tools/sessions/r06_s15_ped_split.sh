# split pooling stage (pool_split.h): parity, kernel times, ped frame rate;
# A/B of the weight-gradient launches' workgroup target
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool_split or pool_weights" 2>&1 | tail -15
echo "== ped_pool_bench (split)"; timeout 300 python tools/ped_pool_bench.py 2>&1 | tail -4
echo "== ped_pool_bench (mlp_debug 8192: LDS-tile kernel)"; PGNN_TUNE=mlp_debug=8192 timeout 300 python tools/ped_pool_bench.py 2>&1 | tail -4
for rep in 1 2; do
for dbg in 0 8192; do
echo "== ped frames/s mlp_debug=$dbg"
timeout 300 python bench.py --config ped_cyl_auto_T3 --no-cpu-baseline --no-roofline --no-secondary --steps 24 --tune mlp_debug=$dbg 2>gpurun_out/s15.err | head -1 | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print(b['value'], b['ms_per_step'])" || tail -3 gpurun_out/s15.err
done; done
for rep in 1 2; do
for t in 768 512 384; do
  echo "== wgrad_wg_target=$t"
  timeout 200 python bench.py --train --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary --tune wgrad_wg_target=$t 2>gpurun_out/abw.err | head -1 | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print(b.get('ms_per_step'), b.get('value'))" || tail -3 gpurun_out/abw.err
done
done
