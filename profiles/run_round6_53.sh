#!/bin/bash
# round 6, call 53: k_pl_lone as committed: the whole -m gpu suite, the extended parity sweeps
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_53; mkdir -p $O
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) | tee $O/tests.log
( timeout 1500 python tests/extended_fuzz.py 150 0 2>&1 | tail -2 ) | tee $O/fuzz_0.txt
( timeout 900 python tests/extended_fuzz.py 70 1000 2>&1 | tail -2 ) | tee $O/fuzz_1000.txt
( timeout 900 python tests/extended_fuzz.py 60 2000 2>&1 | tail -2 ) | tee $O/fuzz_2000.txt
( timeout 900 python tests/extended_fuzz.py 40 3000 2>&1 | tail -2 ) | tee $O/fuzz_3000.txt
