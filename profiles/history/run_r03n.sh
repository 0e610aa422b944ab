#!/bin/bash
# the whole GPU test suite, then the default bench line (what the driver runs)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03n}
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_path"))
for k,v in d.get("also",{}).items():
    if isinstance(v,dict):
        print(k, {x:v[x] for x in ("value","ms_per_step","wall_s","error","skipped") if x in v}, "cpu:", (v.get("cpu_baseline") or {}).get("value") if isinstance(v.get("cpu_baseline"),dict) else v.get("cpu_baseline"))
PY
tail -3 gpurun_out/${TAG}_bench.err
