#!/bin/bash
# do the ranges' tails (the giant cells of the graph / EM kernels) hide under the next range when ranges may overlap?  + taper sweeps
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
run() { env "$@" timeout 300 python bench.py --workload configs2 --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', d['ms_per_step'])"; }
run A=1
run AFQ_RANGE_OVERLAP=1
run AFQ_PUG_TAPER=0.2,0.5,0.78
run AFQ_PUG_TAPER=0.25,0.55,0.8 AFQ_RANGE_OVERLAP=1
run AFQ_PUG_TAPER=0.15,0.4,0.7
run AFQ_PUG_TAPER=0.5,0.8
run A=2
