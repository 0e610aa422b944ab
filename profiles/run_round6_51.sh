#!/bin/bash
# round 6, call 51: the lone vertices over the range's tiles (k_pl_lone: a wave per 256 slots, a staged class finds its partition from its
# UMI) against the wave-per-partition kernel (libafquant_lonepart.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_51; mkdir -p $O
( timeout 1800 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_cli.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in "--workload configs2" "--workload configs2 --na-model tail"; do for lib in "" lonepart "" lonepart; do
  [ -n "$lib" ] && export AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_$lib.so || unset AFQ_LIB_PATH
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print('${lib:-flat}', d['ms_per_step'], 'lone', k['k_p2_lone'], 'graph', k['k_p2_graph'], 'em', k['k_em'])"; done; done | tee $O/configs2.txt
