#!/bin/bash
# round 4: EM enqueued behind the range's kernels (device-side plan): EM tests, all -m gpu EM/parsimony tests, configs2 line + per-kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_em.py -x -q > $O/pytest_em.log 2>&1; echo "pytest em rc=$?"; tail -5 $O/pytest_em.log
timeout 900 python -m pytest tests -m gpu -q -x -k "em or EM or parsimony or boot or dump" > $O/pytest_k.log 2>&1; echo "pytest -k rc=$?"; tail -4 $O/pytest_k.log
PASSES="stats" bash profiles/run_prof.sh r04f --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r04f > $O/summary.txt 2>&1
head -34 $O/summary.txt
timeout 600 python bench.py --workload configs2 --also none --steps 3 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -3 $O/bench_c2.err
python - $O/bench_c2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("configs2", d["value"], d["ms_per_step"])
print(d["roofline"]["all_kernels_ms_per_step"])
print(d["cpu_baseline"].get("em_arithmetic"))
PY
