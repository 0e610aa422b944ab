#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
run() { env "$@" timeout 300 python bench.py --workload configs1 --na-model tail --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', d['ms_per_step'], {k:v for k,v in d['roofline']['all_kernels_ms_per_step'].items() if v>0.2})"; }
run AFQ_DECODE=recs
run AFQ_DECODE=keys
run AFQ_SLAB_CAP=512
run AFQ_SLAB_CAP=448 AFQ_DECODE=recs
