#!/bin/bash
# round 5, call 22: scATAC with the parse of range r+1 next to the sort of range r (own stream), against parse-in-line; 6 / 7 / 8 ranges
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_22; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_atac.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
atac() {
  env AFQ_LIB_PATH=$2 timeout 300 python bench.py --workload atac --steps 5 --warmup 2 --no-cpu-baseline > $O/$1.json 2> $O/$1.err
  python -c "
import json; d=json.load(open('$O/$1.json')); print('$1', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])" 2>&1 | tail -1
}
atac atac $L/libafquant.so
atac atac_pa0 $L/libafquant_pa0.so
atac atac_g1 $L/libafquant_g1.so
atac atac_g2 $L/libafquant_g2.so
atac atac_again $L/libafquant.so
find $O -size +8M -delete
