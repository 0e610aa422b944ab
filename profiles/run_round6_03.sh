#!/bin/bash
# round 6, call 3: + the covers a workgroup per tile (k_pc_cover) - parity of the parsimony tests, then per-kernel times on configs[2]
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_03; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
PASSES="stats" bash profiles/run_prof.sh r6c_configs2 --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r6c_configs2 > $O/r6c_configs2_rocprof.txt 2>&1
head -45 $O/r6c_configs2_rocprof.txt
rm -rf gpurun_out/prof_r6c_configs2
