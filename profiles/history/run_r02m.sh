#!/bin/bash
# round 2: the whole -m gpu suite, the default bench line, and the committed rocprof evidence (default workload + configs[2]).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02m_pytest.log 2>&1
tail -4 gpurun_out/r02m_pytest.log | head -2
( time timeout 900 python bench.py ) > gpurun_out/r02m_bench.json 2> gpurun_out/r02m_bench.err
PASSES="stats sq sq2 fetch write" bash profiles/run_prof.sh r2m --workload configs1 > gpurun_out/r02m_prof.log 2>&1
PASSES="stats sq sq2 fetch write" bash profiles/run_prof.sh r2m_cfg2 --workload configs2 > gpurun_out/r02m_prof_cfg2.log 2>&1
timeout 600 python bench.py --workload atac --steps 3 --warmup 1 > gpurun_out/r02m_bench_atac.json 2> gpurun_out/r02m_bench_atac.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 2 --warmup 1 --cells 2000 --c3-cells 20000 --also configs3 > gpurun_out/r02m_bench_2ranks_shared.json 2> gpurun_out/r02m_bench_2ranks_shared.err
tail -c 300 gpurun_out/r02m_bench_atac.json
