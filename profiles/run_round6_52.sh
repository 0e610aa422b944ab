#!/bin/bash
# round 6, call 52: k_pl_lone with four rows of 64 slots in flight per trip (a 256-slot chunk in one trip; 71 VGPRs, seven waves per SIMD;
# compiled for eight: spills) against two (42 VGPRs, eight waves)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_52; mkdir -p $O
for w in "--workload configs2" "--workload configs2 --na-model tail"; do for lib in "" rows4 rows4w8 "" rows4 rows4w8; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-rows2}', d['ms_per_step'], 'lone', k['k_p2_lone'])"; done; done | tee $O/configs2.txt
