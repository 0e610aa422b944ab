#!/bin/bash
# round 6, call 32: k_p2_search's own-change loop at four vector instructions per change and row (filter words by one exclusive-or
# of byte offsets, hit bits at the deciding bit's position) against the previous commit's library (libafquant_prev.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_32; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in "" prev "" prev; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-new}', d['ms_per_step'], 'search', k['k_p2_search'], 'part', k['k_p2_part'], 'lone', k['k_p2_lone'])"
done | tee $O/configs2.txt
