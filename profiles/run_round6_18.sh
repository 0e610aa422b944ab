#!/bin/bash
# round 6, call 18: the covers' rare gene lists in LDS rows (no scratch in any kernel of the flat graph phase); the tailed model's kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_18; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for W in "configs2" "configs2 --na-model tail"; do
  T=r6p_$(echo $W | tr -d ' -' )
  PASSES="stats" bash profiles/run_prof.sh $T --workload $W > /dev/null 2>&1
  python profiles/summarize.py $T > $O/${T}_rocprof.txt 2>&1
  head -30 $O/${T}_rocprof.txt | cut -c1-100
  tail -3 $O/${T}_rocprof.txt | cut -c1-700
  rm -rf gpurun_out/prof_$T
done
