#!/bin/bash
# full GPU suite (default and capped-builder modes for the graph tests) + the default bench line
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s16
rm -rf $OUT; mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_all.log 2>&1
tail -4 $OUT/pytest_all.log
( PGNN_TUNE="graph_max_wgs=8,graph_lds_pad=32768,ws_reserve=8" timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deferred.py -m gpu -x -q ) > $OUT/pytest_capped.log 2>&1
tail -2 $OUT/pytest_capped.log
( timeout 600 python bench.py --no-live-pmc ) > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); c=d["config"]
print("fps %.1f" % d["value"], "car %.1f" % c["secondary"]["frames_per_sec"], "ped %.1f" % c["secondary_ped"]["frames_per_sec"], "train %.1f" % c["secondary_train"]["training_frames_per_sec"])
print("phase", {k: round(v,3) for k,v in c["phase_ms_frame_seed0"].items()}, "latency", {k: round(v,3) for k,v in c["latency_ms_frame_seed0"].items()}, "overflow", c["capacity_overflow_rebuilds"])
print(c["schedule"]); print(d["roofline_graph"])
PY
