#!/bin/bash
# ranges side by side (AFQ_RANGE_OVERLAP) against one after the other, three runs each
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
run() { local W=$1; shift; env "$@" timeout 300 python bench.py $W --steps 5 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W $*', 'ms_per_step', d['ms_per_step'])"; }
for i in 1 2 3; do run "--workload configs2" A=1; run "--workload configs2" AFQ_RANGE_OVERLAP=1; done
for i in 1 2; do run "--workload configs1" A=1; run "--workload configs1" AFQ_RANGE_OVERLAP=1; done
run "--workload configs2 --na-model tail" A=1; run "--workload configs2 --na-model tail" AFQ_RANGE_OVERLAP=1
