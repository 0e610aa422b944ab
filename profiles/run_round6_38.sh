#!/bin/bash
# round 6, call 38: k_p2_lone<false> compiled for eight waves per SIMD (64 VGPRs, 5 spilled) against seven (70, none)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_38; mkdir -p $O
for lib in "" lone8 "" lone8; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-wpe7}', d['ms_per_step'], 'lone', k['k_p2_lone'])"
done | tee $O/configs2.txt
