mkdir -p gpurun_out
for rep in 1 2; do
for t in 768 512 384; do
  echo "== wgrad_wg_target=$t"
  timeout 200 python bench.py --train --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-secondary --tune wgrad_wg_target=$t 2>gpurun_out/abw.err | python -c "import json,sys; b=json.load(sys.stdin); print(b.get('ms_per_step'), b.get('value'))" || tail -3 gpurun_out/abw.err
done
done
