#!/bin/bash
# round 6, call 43-44: tiled_bitonic_sort_by - partly filled tiles (43), two half-cleaner stages per LDS visit in registers (44):
# for its second tile of 1 616 as for the first): the primitives' check incl. the tiled sort, every suite that sorts through it
# (scATAC, the canonical EM and bootstraps, the one-workgroup parsimony kernel), the scATAC leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_43; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_prims.py tests/test_gpu_atac.py tests/test_gpu_em.py tests/test_gpu_cli.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
( timeout 900 python -m pytest tests/test_gpu_pug.py -x -q -m gpu -k "one-workgroup or handed-back" 2>&1 | tail -3 ) | tee -a $O/tests.log
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also none --workload atac 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"; done | tee $O/atac.txt
