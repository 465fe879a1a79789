for rep in 1 2; do
for lib in "" ab/libnohidden.so; do
echo "== car pool kernel, lib=${lib:-tree}"
PGNN_LIB=${lib:+$PWD/$lib} timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-live-pmc --steps 8 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.readline()); print('pool_us %.1f (frac %.3f)' % (b['roofline_pool']['avg_launch_us'], b['roofline_pool']['frac']))"
echo "== ped pool stage, lib=${lib:-tree}"
PGNN_LIB=${lib:+$PWD/$lib} timeout 300 python tools/ped_pool_bench.py 2>&1 | grep "ped pooling"
done; done
