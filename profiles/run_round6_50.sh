#!/bin/bash
# round 6, call 50: k_p2_search - a vertex that probes its own UMI skips its own table slot before anything else
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_50; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu -k "phase-kernels or random" 2>&1 | tail -3 ) | tee $O/tests.log
for lib in "" prev "" prev; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload configs2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-new}', d['ms_per_step'], 'search', k['k_p2_search'])"
done | tee $O/configs2.txt
