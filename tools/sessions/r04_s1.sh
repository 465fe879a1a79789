#!/bin/bash
# round 4, session 1: new GPU tests of the bench plumbing + where the edge
# kernel's SIMD end-time spread comes from
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "bench_two_ranks" 2>&1 | tail -15 > gpurun_out/r04_s1_tests.log
python tools/ws_timeline.py --balance --cost-model > gpurun_out/r04_s1_timeline.txt 2>&1
python bench.py --no-cpu-baseline --no-live-pmc > gpurun_out/r04_s1_bench.json 2> gpurun_out/r04_s1_bench.err
tail -5 gpurun_out/r04_s1_tests.log; tail -30 gpurun_out/r04_s1_timeline.txt
