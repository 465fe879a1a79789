#!/bin/bash
# round-2 session: where the weights-stationary edge kernel spends its time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "=== timeline default"; timeout 200 python tools/ws_timeline.py 2>&1 | grep -v amdgpu.ids
echo "=== timeline ws_prio=1"; timeout 200 python tools/ws_timeline.py --tune=ws_prio=1 2>&1 | grep -v amdgpu.ids
echo "=== timeline ws_xcds=1"; timeout 200 python tools/ws_timeline.py --tune=ws_xcds=1 2>&1 | grep -v amdgpu.ids
echo "=== PMC"; timeout 600 bash tools/pmc_sq.sh car_600k 2>&1 | grep -v "pool \|amdgpu.ids"
