#!/bin/bash
# round 5, session 18: timing ablations of the f16x2 edge kernel (WRONG results;
# PGNN_F16_ABL bits: 1 no split arithmetic, 2 every row request to row 0,
# 4 fragments read once per tile, 8 no segmented max)
cd "$GRAFT_REPO_ROOT"
for v in "" 1 2 4 8 3 7 15; do
  L=${v:+ab/libf16abl$v.so}
  echo "== abl ${v:-0}"
  PGNN_LIB=$L timeout 200 python tools/bf16x3_bench.py 2>&1 | grep -E "fp16 x2"
done
