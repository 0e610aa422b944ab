#!/bin/bash
# round 6, call 34b: k_p2_check with the flags as atomic ORs again, record offsets only for hashed labels (34: the search flagging its
# candidates itself - graph phase +2.2 ms for the one-vertex components of the vertices that lose all their candidates)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_34; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print(d['ms_per_step'], 'search', k['k_p2_search'], 'part', k['k_p2_part'], 'lone', k['k_p2_lone'], 'graph', k['k_p2_graph'])"
done | tee $O/configs2.txt
PASSES="stats" bash profiles/run_prof.sh t34 --workload configs2 > /dev/null 2>&1; python profiles/summarize.py t34 2>&1 | head -14 | tee $O/prof.txt
