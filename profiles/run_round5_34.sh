#!/bin/bash
# round 5, call 34: the covers sort a staged label's genes in the lane's LDS row, not in an array of the lane's own (scratch memory)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_34; mkdir -p $O
( time timeout 500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x ) > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, lib, then bench flags
  local N=$1 LIB=$2; shift 2
  env AFQ_LIB_PATH=$LIB timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
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
