#!/bin/bash
# big cells on 1024-thread graph workgroups: pug tests (4 routes), configs2 per-launch kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r04o_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04o_pytest.log
PASSES="stats" bash profiles/run_prof.sh r04o --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r04o 2>&1 | head -12
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r04o/stats/stats_results.db')
rows=list(db.execute("select name,start,end from kernels order by start"))
for key in ("k_p2_graph<1024","k_p2_graph<256","k_p2_cover"):
    print(key, [round((e-s)/1e3) for n,s,e in rows if key in n][-6:])
PY
for B in 100000 60000 150000; do AFQ_P2_BIG_READS=$B timeout 300 python bench.py --workload configs2 --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('big>=$B: ms_per_step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"; done
