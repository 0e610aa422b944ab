#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py -x -q -k "not configs3" 2>&1 | tail -4
bash profiles/run_stress.sh 4
timeout 200 python bench.py --workload configs2 --steps 3 --warmup 1 --also none --cpu-seconds 3 > /tmp/o.json 2>/tmp/e.txt
python - <<PY
import json
try:
    d=json.load(open("/tmp/o.json"))
    k=d["roofline"]["all_kernels_ms_per_step"]
    print("configs2:", d["ms_per_step"], {x:k[x] for x in k if k[x]>1}); print(d["cpu_baseline"]["sample"][:80])
except Exception as e: print("fail", e, open("/tmp/e.txt").read()[-400:])
PY
timeout 200 python bench.py --workload configs2 --na-model tail --steps 2 --warmup 1 --also none --cpu-seconds 4 > /tmp/o.json 2>/tmp/e.txt
python - <<PY
import json
try:
    d=json.load(open("/tmp/o.json"))
    k=d["roofline"]["all_kernels_ms_per_step"]
    print("parsimony-em tail:", d["ms_per_step"], {x:k[x] for x in k if k[x]>1}); print(d["cpu_baseline"]["sample"][:80])
except Exception as e: print("fail", e, open("/tmp/e.txt").read()[-400:])
PY
