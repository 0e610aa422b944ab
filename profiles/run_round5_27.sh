#!/bin/bash
# round 5, call 27: what the EM class pass costs without its atomics (timing build, wrong rows on purpose)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_27; mkdir -p $O
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
for m in plain tail; do
  env AFQ_LIB_PATH=$L/libafquant_timing_noadd.so timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none --workload configs2 --na-model $m 2>&1 | grep "^em2 hybrid\|^em2 rounds" > $O/em_$m.txt
  grep hybrid $O/em_$m.txt | head -4; grep "rounds tier=3" $O/em_$m.txt | head -3
done
