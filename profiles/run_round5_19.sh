#!/bin/bash
# round 5, call 19: partitions planned at a mean of at most 96 / 112 / 128 reads against 144 (fewer partitions over 128 reads = fewer four-row sorts)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_19; mkdir -p $O
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, lib, bench flags
  local N=$1 LIB=$2; shift 2
  env AFQ_LIB_PATH=$LIB timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
  python - "$N" "$O/$N.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1], d["ms_per_step"], {a: round(b, 2) for a, b in k.items() if b > 0.3}, "mono", d["retries"].get("cells_through_the_one_workgroup_kernel"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
for v in t96 t112 t128; do one c2_$v $L/libafquant_$v.so --workload configs2; done
one c2_t144 $L/libafquant.so --workload configs2
for v in t96 t112; do one c2t_$v $L/libafquant_$v.so --workload configs2 --na-model tail; done
