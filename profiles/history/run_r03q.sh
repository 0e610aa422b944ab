#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import torch" 2>/dev/null
for RES in table sort; do
AFQ_RESOLVE=$RES timeout 200 python bench.py --na-model tail --steps 5 --warmup 2 --also none --cpu-seconds 3 > /tmp/o.json 2>/tmp/e.txt
python - <<PY
import json
try:
    d=json.load(open("/tmp/o.json"))
    print("cr-like tail, resolve=$RES:", d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("fail", e, open("/tmp/e.txt").read()[-300:])
PY
done
timeout 200 python bench.py --workload configs2 --na-model tail --steps 2 --warmup 1 --also none --cpu-seconds 4 > /tmp/o.json 2>/tmp/e.txt
python - <<PY
import json
try:
    d=json.load(open("/tmp/o.json"))
    print("parsimony-em tail:", d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"]); print(d["cpu_baseline"]["sample"][:80])
except Exception as e: print("fail", e, open("/tmp/e.txt").read()[-400:])
PY
