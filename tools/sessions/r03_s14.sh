#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/s14
for spec in "$@"; do
echo "== $spec"; timeout 300 python tools/pipe_events.py $spec 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s14/run.log | head -3; tail -8 gpurun_out/s14/run.log
done
