#!/bin/bash
# phase clocks of the graph kernel (instrumented build), configs2 one step
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline > gpurun_out/r04z_clocks.txt 2> gpurun_out/r04z_clocks.err
grep "p2 graph" gpurun_out/r04z_clocks.txt | sort -t= -k2 -n -r | awk 'NR%2==1' | head -30
