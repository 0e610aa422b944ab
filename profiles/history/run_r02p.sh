#!/bin/bash
# EM phase clocks (instrumented build)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
AFQ_LIB_PATH=$ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --workload configs2 --steps 1 --warmup 0 --no-cpu-baseline --also none 2>&1 | grep "^em " | cut -c1-300 | sort | uniq | head -60
