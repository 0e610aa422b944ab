#!/bin/bash
# round 5, call 33: EM classes of a run handed their slots with a stride (neighbouring molecules carry the same label: same-address LDS atomics within a wave)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_33; mkdir -p $O
( time timeout 400 python -m pytest tests/test_gpu_em.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -6 $O/tests.log | grep -v "^$" | cut -c1-300
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
one c2 $L/libafquant.so --workload configs2
one c2t $L/libafquant.so --workload configs2 --na-model tail
for m in tail; do
  env AFQ_LIB_PATH=$L/libafquant_timing.so timeout 100 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none --workload configs2 --na-model $m 2>&1 | grep "^em2 hybrid\|^em2 rounds tier=3" > $O/em_$m.txt
  grep hybrid $O/em_$m.txt | head -6;  grep "rounds tier=3" $O/em_$m.txt | head -2
done
