#!/bin/bash
# round 4, session 20: training step with / without the balanced partition
cd "$GRAFT_REPO_ROOT"
for bal in 1 0 1; do
  echo "== ws_balance=$bal"
  python bench.py --train --steps 24 --warmup 8 --frames 4 --tune ws_balance=$bal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
