#!/bin/bash
# round 6, call 27: the extended parity sweeps on the flat graph phase with the 24-word pool: parsimony (seeds 0..), every resolution (1000..), the tailed model (2000.., 3000..)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_27; mkdir -p $O
( timeout 1500 python tests/extended_fuzz.py 150 0 2>&1 | tail -6 ) | tee $O/fuzz_0.txt
( timeout 900 python tests/extended_fuzz.py 70 1000 2>&1 | tail -4 ) | tee $O/fuzz_1000.txt
( timeout 900 python tests/extended_fuzz.py 60 2000 2>&1 | tail -4 ) | tee $O/fuzz_2000.txt
( timeout 900 python tests/extended_fuzz.py 40 3000 2>&1 | tail -4 ) | tee $O/fuzz_3000.txt
