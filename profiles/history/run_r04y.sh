#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
run() { env "$@" timeout 300 python bench.py --workload configs2 --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', d['ms_per_step'], 'graph', d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"; }
run AFQ_P2_BIG_READS=25000
run AFQ_P2_BIG_READS=18000
run AFQ_P2_BIG_READS=12000
run AFQ_P2_BIG_READS=6000
run AFQ_P2_BIG_READS=1
