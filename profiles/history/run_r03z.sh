#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z; mkdir -p $O
python -c "import torch" 2>/dev/null
B="--steps 10 --warmup 3 --also none --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], round(sum(k.values()),2), {x:round(k[x],2) for x in k if k[x]>0.3})
except Exception as e: print(sys.argv[1], "fail", e)
PY
}
run() { name=$1; shift; env "$@" timeout 120 python bench.py $B > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
run base A=1
run base2 A=1
run overlap AFQ_RANGE_OVERLAP=1
run t4 AFQ_CR_TAPER=0.35,0.65,0.88
run t4_overlap AFQ_CR_TAPER=0.35,0.65,0.88 AFQ_RANGE_OVERLAP=1
run t3 AFQ_CR_TAPER=0.45,0.82
run t6_overlap AFQ_CR_TAPER=0.25,0.5,0.7,0.86 AFQ_RANGE_OVERLAP=1
run nopipe AFQ_NO_PIPELINE=1
AFQ_HOST_TIMING=1 timeout 120 python bench.py --steps 2 --warmup 1 --also none --no-cpu-baseline 2>&1 | grep -i "run:\|finish\|collect\|submit" | tail -40 > $O/host_timing.txt; tail -30 $O/host_timing.txt
