#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline > gpurun_out/r04d_clocks.txt 2> gpurun_out/r04d_clocks.err
grep "em2 blk" gpurun_out/r04d_clocks.txt | wc -l
