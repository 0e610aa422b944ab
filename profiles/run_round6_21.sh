#!/bin/bash
# round 6, call 21: what the covers' output reservations (one to three returned atomics per wave and batch) cost: a variant without them (wrong rows)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_21; mkdir -p $O
export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_noout.so
PASSES="stats" bash profiles/run_prof.sh r6r_configs2 --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r6r_configs2 > $O/r6r_configs2_rocprof.txt 2>&1
grep -E "k_pc_|k_pf_|k_p2_tied" $O/r6r_configs2_rocprof.txt | head -20
tail -3 gpurun_out/prof_r6r_configs2/stats.err | cut -c1-300
rm -rf gpurun_out/prof_r6r_configs2
