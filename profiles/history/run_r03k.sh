#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
for G in 0 2048 8192; do
AFQ_P2_GRID=$G timeout 100 python bench.py --workload configs2 --steps 3 --warmup 1 --also none --no-cpu-baseline > /tmp/o.json 2>/dev/null
python - <<PY
import json
d=json.load(open("/tmp/o.json"))
k=d["roofline"]["all_kernels_ms_per_step"]
print("grid $G", d["ms_per_step"], {x:k[x] for x in k if x.startswith("k_p2") or x=="k_em"})
PY
done
