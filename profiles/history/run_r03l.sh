#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03l}
timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py -x -q -k "not configs3" 2>&1 | tail -6
timeout 200 python bench.py --workload configs2 --steps 3 --warmup 1 --also none --cpu-seconds 4 > gpurun_out/${TAG}_cfg2.json 2> gpurun_out/${TAG}_cfg2.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_cfg2.json"))
k=d["roofline"]["all_kernels_ms_per_step"]
print(d["value"], d["ms_per_step"], {x:k[x] for x in k if k[x] > 1})
print((d.get("cpu_baseline") or {}).get("sample","")[:90])
PY
tail -2 gpurun_out/${TAG}_cfg2.err
