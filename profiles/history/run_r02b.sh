#!/bin/bash
# round 2, second GPU call: the rest of the -m gpu suite (no -x), and the per-phase device clocks of the parsimony / EM
# kernels on configs[2] (instrumented build, AFQ_LIB_PATH).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02b_pytest.log 2>&1
tail -8 gpurun_out/r02b_pytest.log
AFQ_LIB_PATH=$ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --workload configs2 --steps 1 --warmup 0 --no-cpu-baseline --also none > gpurun_out/r02b_timing_cfg2.log 2>&1
grep -c "pug cell" gpurun_out/r02b_timing_cfg2.log
