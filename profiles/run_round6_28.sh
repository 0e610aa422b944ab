#!/bin/bash
# round 6, call 28: three of a range's 5 us launches folded into their neighbours (the proof's last step into the fix-up decode, the
# big overflow buckets into k_resolve_mid, k_pack_small into k_row_ptr): the cr-like suites, the headline
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_28; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_multi.py tests/test_gpu_cli.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"; done | tee $O/headline.txt
