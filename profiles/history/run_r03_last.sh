#!/bin/bash
# the round's last GPU call: the whole -m gpu suite and the default bench line on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_last; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("label_rehashes"))
for k,v in d["also"].items(): print(k, v.get("value"), v.get("ms_per_step", v.get("wall_s")), v.get("error", ""), v.get("slowdown_per_input_byte_vs_plain",""))
PY
