#!/bin/bash
# graph kernel split into graph + cover: pug tests, configs2 per-kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r04l_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04l_pytest.log
PASSES="stats" bash profiles/run_prof.sh r04l --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r04l 2>&1 | head -16
for W in 4 8; do AFQ_P2_COVER_WGS=$W timeout 300 python bench.py --workload configs2 --steps 3 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cover wgs $W: ms_per_step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"; done
