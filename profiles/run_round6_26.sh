#!/bin/bash
# round 6, call 26: how much of the parsimony pool a range uses (plain and tailed model)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_26; mkdir -p $O
for W in "" "--na-model tail"; do
AFQ_TEST_PF_STATS=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none --workload configs2 $W 2>&1 | grep "flat graph" | head -3 | tee -a $O/sizes.txt
done
