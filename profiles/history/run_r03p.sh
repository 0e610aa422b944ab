#!/bin/bash
# new tests of this batch of changes, then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03p}
timeout 900 python -m pytest tests/test_gpu_pug.py tests/test_gpu_multi.py -x -q 2>&1 | tail -12
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_path"))
for k,v in d.get("also",{}).items():
    if isinstance(v,dict):
        print(k, {x:v[x] for x in ("value","ms_per_step","wall_s","error","skipped","slowdown_per_input_byte_vs_plain") if x in v}, "cpu:", (v.get("cpu_baseline") or {}).get("value") if isinstance(v.get("cpu_baseline"),dict) else v.get("cpu_baseline"))
c2=d["also"]["configs2"]["cpu_baseline"]
print({k:c2[k] for k in c2 if k not in ("sample",)})
PY
tail -3 gpurun_out/${TAG}_bench.err
