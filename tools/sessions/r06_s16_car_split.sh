# car's pooling stage on the two-launch form (pool_split_car=1) vs pool_ws.h:
# parity, standalone stage time, frame rate; per-kernel split of the ped stage;
# secondary_train at 24 vs 64 timed steps
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool_split or pool_weights" 2>&1 | tail -5
for rep in 1 2; do
for t in 0 1; do
  echo "== pool_split_car=$t"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 32 --tune pool_split_car=$t  # (tunable of the session's tree; the form was not kept) 2>gpurun_out/s16.err \
    | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print('frames/s %.1f edge_us %.1f (frac %.3f) pool_us %.1f (frac %.3f)' % (b['value'], b['roofline_mfma']['avg_launch_us'], b['roofline_mfma']['frac'], b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))" || tail -5 gpurun_out/s16.err
done; done
echo "== ped pooling stage under rocprofv3"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/s16_prof -o run -- python $OLDPWD/tools/ped_pool_bench.py > $OLDPWD/gpurun_out/s16_prof.log 2>&1)
db=$(find gpurun_out/s16_prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" gpurun_out/s16_ped_pool_kernel_stats > /dev/null; head -12 gpurun_out/s16_ped_pool_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/s16_prof
echo "== car pooling stage (split) under rocprofv3"
(cd /tmp && PGNN_TUNE=pool_split_car=1 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/s16_prof -o run -- python $OLDPWD/bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 4 --frames 1 --no-pipeline > $OLDPWD/gpurun_out/s16_prof2.log 2>&1)
db=$(find gpurun_out/s16_prof -name "*.db" | head -1)
python tools/prof_summary.py "$db" gpurun_out/s16_car_split_kernel_stats > /dev/null; head -10 gpurun_out/s16_car_split_kernel_stats.md | cut -c1-150
rm -rf gpurun_out/s16_prof
for st in 24 64 24 64; do
  echo "== train --frames 4 --steps $st"
  timeout 200 python bench.py --train --frames 4 --steps $st --warmup 8 --no-live-pmc 2>gpurun_out/s16t.err | head -1 | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print(b.get('ms_per_step'), b.get('value'))" || tail -3 gpurun_out/s16t.err
done
