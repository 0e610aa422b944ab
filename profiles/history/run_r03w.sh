#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03w; mkdir -p $O
python -c "import torch" 2>/dev/null
B="--steps 3 --warmup 1 --also none --no-cpu-baseline --workload configs2"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], {x:round(k[x],1) for x in k if k[x]>3})
except Exception as e: print(sys.argv[1], "fail", e)
PY
}
run() { name=$1; shift; env "$@" timeout 120 python bench.py $B > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
run base A=1
run overlap AFQ_RANGE_OVERLAP=1
run t4 AFQ_PUG_TAPER=0.30,0.60,0.85
run t4_overlap AFQ_PUG_TAPER=0.30,0.60,0.85 AFQ_RANGE_OVERLAP=1
run t5_overlap AFQ_PUG_TAPER=0.25,0.5,0.72,0.9 AFQ_RANGE_OVERLAP=1
run t3b_overlap AFQ_PUG_TAPER=0.36,0.72 AFQ_RANGE_OVERLAP=1
run t6 AFQ_PUG_TAPER=0.2,0.4,0.6,0.8
