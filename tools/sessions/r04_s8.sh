#!/bin/bash
# round 4, session 8: what a K-row layer pass waits for -- ablations
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "" noload sameq; do
  if [ -z "$v" ]; then python tools/krow_timeline.py; else PGNN_LIB=$PWD/ab/lib$v.so python tools/krow_timeline.py; fi
done > gpurun_out/r04_s8_krow_abl.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_s8_krow_abl.txt
