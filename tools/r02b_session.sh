#!/bin/bash
# Round-2 end-state evidence (via gpurun): PMC of the roofline kernel first (so
# that bench.py finds its traffic sidecar), whole GPU suite, smoke, bench lines,
# rocprofv3 kernel stats, SQ counters of the MFMA kernels.  -> gpurun_out/r02/
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out/r02
bash tools/pmc_scatter.sh > gpurun_out/r02/pmc_scatter.log 2>&1
python tools/pmc_scatter_json.py gpurun_out profiles/r02_pmc_scatter_max.json > gpurun_out/r02/pmc_scatter_max.json 2>> gpurun_out/r02/pmc_scatter.log
rm -rf gpurun_out/pmc2_scatter_FETCH_SIZE gpurun_out/pmc2_scatter_WRITE_SIZE
bash tools/r02_session.sh tests bench prof
timeout 600 bash tools/pmc_sq.sh car_600k > gpurun_out/r02/pmc_sq.log 2>&1
cp gpurun_out/pmc_sq_car_600k.txt gpurun_out/r02/ 2>/dev/null
cat gpurun_out/r02/pmc_scatter_max.json
