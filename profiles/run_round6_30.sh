#!/bin/bash
# round 6, call 30: two new parity tests of the flat build (a component of every size on both sides of each list boundary;
# edge directions from read counts beyond the flag byte's 127), each through the seven routes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_30; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py -q -m gpu -k "boundary or beyond_the_flag" 2>&1 | tail -30 ) > $O/tests.log 2>&1; tail -30 $O/tests.log
