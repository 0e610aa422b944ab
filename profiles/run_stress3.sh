#!/bin/bash
# repeat the configs2 bench N times and print where the time went (an intermittent 1.2 s step was seen in run_r03_final.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-6}
python -c "import torch" 2>/dev/null
for T in $(seq 1 $N); do
  AFQ_HOST_TIMING=1 timeout 40 python bench.py --workload configs2 --steps 2 --warmup 1 --also none --no-cpu-baseline > /tmp/o.txt 2> /tmp/e.txt
  rc=$?
  python - <<PY
import json
try:
    d=json.loads([l for l in open("/tmp/o.txt") if l.startswith("{")][-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    print("try $T rc=$rc", d["ms_per_step"], d.get("label_rehashes"), {x:round(k[x],1) for x in k if k[x]>3})
except Exception as e: print("try $T rc=$rc fail", e)
PY
  grep -E "finish: wait|run: plan|run: uploads|submit|collect" /tmp/e.txt | awk '{a[$3" "$4" "$5]+=$(NF-1)} END {for (k in a) printf "   %s %.1f ms;", k, a[k]; print ""}'
done
