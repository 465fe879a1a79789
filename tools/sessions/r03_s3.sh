#!/bin/bash
# Round-3 GPU session 3: capacity form (no host read in graph build / model):
# its tests, then the bench line with and without it on the same box.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s3
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_deferred.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
if [ "${1:-}" = "full" ]; then
( time timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_deferred.py ) > $OUT/pytest_all.log 2>&1
tail -5 $OUT/pytest_all.log
fi
for mode in "" "--host-sized"; do
  tag=cap; [ -n "$mode" ] && tag=host
  ( timeout 600 python bench.py --no-cpu-baseline --no-live-pmc $mode ) > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  tail -3 $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$tag.json"))
    c=d["config"]
    print("$tag", "fps %.1f" % d["value"], "car %.1f" % c["secondary"]["frames_per_sec"], "ped %.1f" % c["secondary_ped"]["frames_per_sec"], "train %.1f" % c["secondary_train"]["training_frames_per_sec"])
    print("   phase", {k: round(v,3) for k,v in c["phase_ms_frame_seed0"].items()}, "latency", {k: round(v,3) for k,v in c["latency_ms_frame_seed0"].items()}, "overflow", c["capacity_overflow_rebuilds"])
except Exception as e:
    print("$tag failed", e)
PY
done
( timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-roofline --no-secondary --no-pipeline --frames 1 --steps 20 ) > $OUT/bench_seq_cap.json 2>/dev/null
( timeout 300 python bench.py --no-cpu-baseline --no-live-pmc --no-roofline --no-secondary --no-pipeline --frames 1 --steps 20 --host-sized ) > $OUT/bench_seq_host.json 2>/dev/null
python - <<PY
import json
for t in ("cap","host"):
    try:
        d=json.load(open("$OUT/bench_seq_%s.json" % t)); print("sequential seed0", t, "%.3f ms/frame" % d["ms_per_step"])
    except Exception as e: print(t, "failed", e)
PY
