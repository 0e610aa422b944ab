#!/bin/bash
# round 4: the EM tests, the per-block timeline of the rounds kernels (instrumented build), per-kernel times of configs2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_em.py -x -q > $O/pytest_em.log 2>&1; echo "pytest em rc=$?"; tail -5 $O/pytest_em.log
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline > $O/clocks.txt 2> $O/clocks.err
PASSES="stats" bash profiles/run_prof.sh r04e --workload configs2 > /dev/null 2>&1
python profiles/summarize.py r04e > $O/summary.txt 2>&1
head -34 $O/summary.txt
