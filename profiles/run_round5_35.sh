#!/bin/bash
# round 5, call 35: k_resolve's merge of a UMI with more than three genes without the callback (its aggregates lived in scratch memory)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_35; mkdir -p $O
( time timeout 500 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_fuzz.py -m gpu -q -x ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
for i in 1 2; do
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also none > $O/c1_$i.json 2> $O/c1_$i.err
python -c "
import json; d=json.load(open('$O/c1_$i.json')); print('c1', d['ms_per_step'], {a: round(b, 2) for a, b in d['roofline']['all_kernels_ms_per_step'].items()})"
done
