#!/bin/bash
# round 5, session 17: the frame loop with edge_arith (run.py mirror), then the
# whole GPU suite on the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
O=gpurun_out/r05/s17
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -m gpu -p no:cacheprovider > $O.tests1.log 2>&1
echo "TESTS1 rc=$? $(tail -1 $O.tests1.log)"
grep -E "^(FAILED|ERROR)|Error" $O.tests1.log | head
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O.tests2.log 2>&1
echo "TESTS2 rc=$? $(tail -1 $O.tests2.log)"
grep -E "^(FAILED|ERROR)" $O.tests2.log | head
