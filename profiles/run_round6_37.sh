#!/bin/bash
# round 6, call 37: k_p2_search (54 VGPRs since the check left it) fetching 6 / 12 foreign partitions together instead of 4
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_37; mkdir -p $O
for lib in "" fb6 fb12 "" fb6 fb12; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-fb4}', d['ms_per_step'], 'search', k['k_p2_search'])"
done | tee $O/configs2.txt
