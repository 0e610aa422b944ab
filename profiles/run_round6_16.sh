#!/bin/bash
# round 6, call 16: roots skip k_pf_root's atomic; which cells get the 1024-thread instance of k_p2_tied<., true>
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_16; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
PASSES="stats" bash profiles/run_prof.sh r6o_configs2 --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r6o_configs2 > $O/r6o_configs2_rocprof.txt 2>&1
head -45 $O/r6o_configs2_rocprof.txt | grep -E "k_pf|k_pc|k_pt|k_p2_tied|k_p2_cover|k_p2_graph|bench.py"
tail -4 $O/r6o_configs2_rocprof.txt | cut -c1-600
rm -rf gpurun_out/prof_r6o_configs2
for BR in 15000 40000 80000 1000000; do
  echo "== AFQ_TEST_P2_BIG_READS=$BR" | tee -a $O/big_reads.txt
  AFQ_TEST_P2_BIG_READS=$BR python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])" | tee -a $O/big_reads.txt
done
