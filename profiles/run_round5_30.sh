#!/bin/bash
# round 5, call 30: thread 0's clock per level of the hybrid EM's class pass (timing build)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_30; mkdir -p $O
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
for m in plain tail; do
  env AFQ_LIB_PATH=$L/libafquant_timing.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none --workload configs2 --na-model $m 2>&1 | grep "^em2 hybrid" > $O/em_$m.txt
  head -8 $O/em_$m.txt
done
