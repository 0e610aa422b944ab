#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, the default bench line with its legs, and the configs[2] evidence
# (per-kernel stats + SQ counters + FETCH/WRITE passes for k_pug_cell / k_em).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02a_pytest.log 2>&1
tail -5 gpurun_out/r02a_pytest.log
( time AFQ_HOST_TIMING=1 timeout 900 python bench.py ) > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -c 600 gpurun_out/r02a_bench.err
PASSES="stats sq sq2 fetch write" bash profiles/run_prof.sh r2a_cfg2 --workload configs2 > gpurun_out/r02a_prof_cfg2.log 2>&1
PASSES="stats fetch write" bash profiles/run_prof.sh r2a --workload configs1 > gpurun_out/r02a_prof.log 2>&1
nproc; free -g | head -2
