#!/bin/bash
# round 5, call 23: the EM's class pass with every load unconditional (no load under a divergent branch, no flat loads / atomics in the hybrid kernel); scATAC in eight ranges
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_23; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_atac.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, lib, then bench flags
  local N=$1 LIB=$2; shift 2
  env AFQ_LIB_PATH=$LIB timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
  python - "$N" "$O/$N.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1], d["ms_per_step"], {a: round(b, 2) for a, b in k.items() if b > 0.3})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
one c2 $L/libafquant.so --workload configs2
one c2t $L/libafquant.so --workload configs2 --na-model tail
timeout 300 python bench.py --workload atac --steps 5 --warmup 2 --no-cpu-baseline > $O/atac.json 2> $O/atac.err; python -c "
import json; d=json.load(open('$O/atac.json')); print('atac', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
find $O -size +8M -delete
