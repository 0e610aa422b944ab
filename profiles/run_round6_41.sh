#!/bin/bash
# round 6, call 41: k_p2_check with 1 / 2 / 4 candidates per thread and trip
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_41; mkdir -p $O
for lib in "" chk4 chk1; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  PASSES="stats" bash profiles/run_prof.sh t41 --workload configs2 > /dev/null 2>&1; echo "== ${lib:-chk2}"; python profiles/summarize.py t41 2>&1 | grep "k_p2_check\|k_p2_search \|ms_per_step"
done | tee $O/prof.txt
