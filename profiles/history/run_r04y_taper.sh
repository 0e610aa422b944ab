#!/bin/bash
# lab note: the cr-like range taper (AFQ_CR_TAPER = cumulative work fractions at which the batch is cut): ms per step of the headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/taper
python -c "import torch" 2>/dev/null
for t in default 0.40,0.68,0.85,0.95 0.45,0.72,0.88,0.96 0.35,0.62,0.81,0.93,0.98 0.50,0.77,0.91,0.97 0.33,0.60,0.80,0.92,0.97; do
  if [ $t = default ]; then unset AFQ_CR_TAPER; else export AFQ_CR_TAPER=$t; fi
  timeout 200 python bench.py --also none --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/taper/$t.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/taper/$t.json').read().strip().splitlines()[-1]); print('$t', d['ms_per_step'])
PY
done
