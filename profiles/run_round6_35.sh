#!/bin/bash
# round 6, call 35: what k_p2_check's 1.07 ms are - the flags as a byte read + byte write (rmw), no flags at all (noflag: wrong results, timing only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_35; mkdir -p $O
for lib in "" rmw noflag; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  PASSES="stats" bash profiles/run_prof.sh t35 --workload configs2 > /dev/null 2>&1; echo "== ${lib:-atomics}"; python profiles/summarize.py t35 2>&1 | grep "k_p2_check\|k_p2_search \|k_pf_union\|k_pf_count\|ms_per_step"
done | tee $O/prof.txt
