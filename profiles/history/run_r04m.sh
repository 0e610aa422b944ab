#!/bin/bash
# k_p2_graph variants: LDS table size x waves per SIMD (spills against occupancy), AFQ_P2_GRAPH_WGS to match
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
for V in "4096 3 4" "2048 4 4" "2048 4 6" "2048 5 5" "2048 5 8" "2048 6 6" "2048 6 8"; do set -- $V
  AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_g$1_$2.so AFQ_P2_GRAPH_WGS=$3 AFQ_P2_COVER_WGS=8 timeout 300 python bench.py --workload configs2 --steps 3 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gtab $1 wpe $2 wgs $3: ms_per_step', d['ms_per_step'], 'graph+cover', d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"
done
