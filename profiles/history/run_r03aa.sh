#!/bin/bash
# round 3: knobs on the final code - k_cell_hist LDS per pass, graph workgroups per CU, planned partition size
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03aa; mkdir -p $O
python -c "import torch" 2>/dev/null
B="--steps 4 --warmup 1 --also none --no-cpu-baseline --workload configs2"
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
run hist16k AFQ_HIST_WORDS=16384
run hist8k AFQ_HIST_WORDS=8192
run g3 AFQ_P2_GRAPH_WGS=3
run g8 AFQ_P2_GRAPH_WGS=8
run t96 AFQ_P2_TARGET=96
run t128 AFQ_P2_TARGET=128
run t224 AFQ_P2_TARGET=224
run base2 A=1
