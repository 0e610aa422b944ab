#!/bin/bash
# round 5, call 36: the EM's streamed class pass with a two-word form for labels of one or two words (eight classes per thread)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_36; mkdir -p $O
true
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, lib, then bench flags
  local N=$1 LIB=$2; shift 2
  env AFQ_LIB_PATH=$LIB timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
  python - "$N" "$O/$N.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1], d["ms_per_step"], {a: round(b, 2) for a, b in k.items() if a in ("k_em",)})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for v in _nopairs "" _nopairs ""; do
  one c2$v $L/libafquant$v.so --workload configs2
  one c2t$v $L/libafquant$v.so --workload configs2 --na-model tail
done
