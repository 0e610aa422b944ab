#!/bin/bash
# round 6, call 14: where k_p2_tied spends a cell's time (the timing build's per-phase device clocks)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_14; mkdir -p $O
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none --workload configs2 2>&1 | grep "p2 tied" | sort -t= -k2 -n | awk 'NR%6==1' | head -40 | tee $O/tied_phases.txt
