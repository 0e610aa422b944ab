#!/bin/bash
# last check of the round on the final code: the whole -m gpu suite, smoke(), the headline and configs[2] numbers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_check; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for w in configs1 configs2; do
  timeout 120 python bench.py --workload $w --steps 5 --warmup 2 --also none --no-cpu-baseline > $O/$w.json 2> $O/$w.err
  python - $O/$w.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], d.get("label_rehashes"), {x:round(k[x],2) for x in k if k[x]>0.5})
PY
done
bash profiles/run_stress.sh 4 2>&1 | tail -4
