# one frame's overlapped graph build alone, per-launch timeline (which kernels
# of the build overlap; what the GNN's first stage waits for)
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$PWD
rm -rf gpurun_out/s20_prof
(cd /tmp && rocprofv3 --kernel-trace -d $ROOT/gpurun_out/s20_prof -o run -- python $ROOT/tools/build_trace.py > $ROOT/gpurun_out/s20_build.log 2>&1)
cat gpurun_out/s20_build.log | grep -v amdgpu | tail -14
db=$(find gpurun_out/s20_prof -name "*.db" | head -1)
python tools/trace_dump.py "$db" --last-ms 1.0 --out gpurun_out/s20_build_trace.txt
rm -rf gpurun_out/s20_prof
cat gpurun_out/s20_build_trace.txt | cut -c1-110 | head -90
