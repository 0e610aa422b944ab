#!/bin/bash
# round 6, call 17: the whole -m gpu suite and the default bench line on the flat graph phase
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_17; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/tests.log 2>&1; tail -6 $O/tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json; echo; tail -3 $O/bench.err
