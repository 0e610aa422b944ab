#!/bin/bash
# round 6, call 10: which cells get the 1024-thread instance of k_p2_tied now that the graph build is range-wide (AFQ_TEST_P2_BIG_READS)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_10; mkdir -p $O
for BR in 15000 30000 60000 120000 1000000; do
  echo "== AFQ_TEST_P2_BIG_READS=$BR" | tee -a $O/big_reads.txt
  AFQ_TEST_P2_BIG_READS=$BR python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])" | tee -a $O/big_reads.txt
done
