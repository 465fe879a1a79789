for rep in 1 2; do
for lib in ab/libpoolh8.so "" ab/libpoolh16.so; do  # (tree = 12 waves since this session; variants by tools/build_variant.py poolhN gnn.hip -DPGNN_POOLH_WAVES=N)
echo "== ped pool stage, lib=${lib:-tree (8 waves)}"
PGNN_LIB=${lib:+$PWD/$lib} timeout 300 python tools/ped_pool_bench.py 2>&1 | grep "ped pooling"
done; done
PGNN_LIB=$PWD/ab/libpoolh12.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pool_split" 2>&1 | tail -2
