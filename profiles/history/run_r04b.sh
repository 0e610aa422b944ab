#!/bin/bash
# round 4: per-kernel times of configs2 with the new EM (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
PASSES="stats" bash profiles/run_prof.sh r04b --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r04b > gpurun_out/prof_r04b/summary.txt 2>&1
head -40 gpurun_out/prof_r04b/summary.txt
