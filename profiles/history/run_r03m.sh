#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03m}
python -c "import torch" 2>/dev/null
timeout 200 python bench.py --workload configs2 --steps 3 --warmup 1 --also none --cpu-seconds 3 > gpurun_out/${TAG}_cfg2.json 2> gpurun_out/${TAG}_cfg2.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_cfg2.json"))
k=d["roofline"]["all_kernels_ms_per_step"]
print(d["value"], d["ms_per_step"], {x:k[x] for x in k if k[x] > 1})
print((d.get("cpu_baseline") or {}).get("sample","")[:90])
PY
tail -2 gpurun_out/${TAG}_cfg2.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --workload configs2 --steps 2 --warmup 1 --also none --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.err
cd $GRAFT_REPO_ROOT
python profiles/summarize_db.py gpurun_out/prof_${TAG}/cfg2_results.db | head -12
bash profiles/run_r03c.sh $TAG | grep -v "^p2 graph cell R=[0-9]\{4\} " | head -12
