#!/bin/bash
# round 6, call 33: the label test of a found pair out of k_p2_search into k_p2_check (a thread per candidate over the range);
# variant wg8: eight search workgroups per CU with a 2^11-bit filter
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_33; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in "" wg8 "" wg8; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-new}', d['ms_per_step'], 'search', k['k_p2_search'], 'part', k['k_p2_part'], 'lone', k['k_p2_lone'], 'graph', k['k_p2_graph'])"
done | tee $O/configs2.txt
