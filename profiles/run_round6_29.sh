#!/bin/bash
# round 6, call 29: k_p2_lone without scratch (the genes of a label of more than 64 refs in an LDS row, one lane at a time, behind the
# rows' loop): the parsimony suites, configs[2] plain and tailed
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_29; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in "--workload configs2" "--workload configs2 --na-model tail"; do for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"; done; done | tee $O/configs2.txt
