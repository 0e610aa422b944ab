#!/bin/bash
# phase clocks of the parsimony kernels (instrumented build, AFQ_LIB_PATH): configs2, one step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-r03c}
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline > gpurun_out/${TAG}_clocks.txt 2> gpurun_out/${TAG}_clocks.err
grep "p2 graph" gpurun_out/${TAG}_clocks.txt | sort -t= -k2 -n -r | head -60
