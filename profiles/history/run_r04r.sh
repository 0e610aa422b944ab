#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
PASSES="stats" bash profiles/run_prof.sh r04r_c2t --workload configs2 --na-model tail > /dev/null 2>&1
python profiles/summarize.py r04r_c2t 2>&1 | head -30
python - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_r04r_c2t/stats/stats_results.db')
rows=list(db.execute("select name,start,end from kernels order by start"))
for key in ("k_p2_graph<1024","k_p2_graph<256","k_p2_cover","k_pug_cell","k_em2_rounds_hybrid","k_em2_rounds<1024, 0","k_em2_rounds<1024, 1","k_em2_rounds<512","k_em2_setup","k_p2_lone","k_p2_search"):
    print(key, [round((e-s)/1e3) for n,s,e in rows if key in n][-3:])
PY
