#!/bin/bash
# round 4: the default bench line (every leg) after the bench.py fixes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_step"], d["roofline"]["traffic"], d.get("retries"))
for k,v in d["also"].items():
    r=v.get("roofline") or {}
    print(k, v.get("value"), v.get("ms_per_step", v.get("wall_s")), v.get("error", ""), v.get("slowdown_per_input_byte_vs_plain",""), r.get("kernel"), r.get("frac"), r.get("frac_step"), r.get("traffic"), (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("em_arithmetic"))
PY
