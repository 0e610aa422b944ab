#!/bin/bash
# round 6, call 49: a single-ref label's signature field holds the ref itself (ref ids within 19 bits): two single-ref labels are compared
# exactly by the search - fewer candidates for k_p2_check.  Parsimony suites, EM, configs[2] plain and tailed, candidate counts
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_49; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu -k "phase-kernels or random" 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in "--workload configs2" "--workload configs2 --na-model tail"; do for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print(d['ms_per_step'], 'search', k['k_p2_search'], 'part', k['k_p2_part'], 'lone', k['k_p2_lone'], 'graph', k['k_p2_graph'], 'em', k['k_em'])"; done; done | tee $O/configs2.txt
