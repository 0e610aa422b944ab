#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03e}
timeout 420 python -m pytest tests/test_gpu_pug.py -x -q 2>&1 | tail -8 > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_pytest.log
for V in direct; do
AFQ_P2_SCATTER=$V timeout 300 python bench.py --workload configs2 --steps 2 --warmup 1 --also none --cpu-seconds 3 > gpurun_out/${TAG}_cfg2_$V.json 2> gpurun_out/${TAG}_cfg2.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_cfg2_$V.json"))
print("$V", d["value"], d["ms_per_step"], json.dumps(d["roofline"]["all_kernels_ms_per_step"]))
PY
done
tail -3 gpurun_out/${TAG}_cfg2.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -o cfg2 -- python $GRAFT_REPO_ROOT/bench.py --workload configs2 --steps 2 --warmup 1 --also none --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.err
cd $GRAFT_REPO_ROOT
python profiles/summarize_db.py gpurun_out/prof_${TAG}/cfg2_results.db | head -14
bash profiles/run_r03c.sh $TAG | head -24
