#!/bin/bash
# late round 4: the default bench line (every leg) with the new defaults: six geometric ranges, 8 slabs per decoder wave, non-temporal key loads in the resolve
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ae; mkdir -p $O
python -c "import torch" 2>/dev/null
( time timeout 500 python bench.py ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(name, x):
    r = x.get("roofline") or {}
    print(f"{name:16s} {x.get('value')} {x.get('unit')}  {x.get('ms_per_step')} ms  frac {r.get('frac')} ({r.get('kernel')}) frac_step {r.get('frac_step')}  kernels {r.get('all_kernels_ms_per_step')}")
show("headline", d)
for k, v in d["also"].items():
    if isinstance(v, dict) and "ms_per_step" in v: show(k, v)
    else: print(k, json.dumps(v)[:300])
print("cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
PY
