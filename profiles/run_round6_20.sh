#!/bin/bash
# round 6, call 20: k_p2_tied streams the record offsets with the keys
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_20; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
PASSES="stats" bash profiles/run_prof.sh r6q_configs2 --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r6q_configs2 > $O/r6q_configs2_rocprof.txt 2>&1
head -45 $O/r6q_configs2_rocprof.txt | grep -E "k_pf|k_pc|k_pt|k_p2_tied|k_p2_cover|k_p2_graph|bench.py"
tail -4 $O/r6q_configs2_rocprof.txt | cut -c1-600
rm -rf gpurun_out/prof_r6q_configs2
