#!/bin/bash
# round 5, call 3: partition kernels with their metadata fetched 64 partitions ahead and their first loads one partition ahead;
# search variants (workgroups per CU, foreign partitions fetched together); k_resolve with its sort fallback as a call.
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_03; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, lib, extra env..., then bench flags after --
  local N=$1 LIB=$2; shift 2
  local ENVS=()
  while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; shift
  env AFQ_LIB_PATH=$LIB "${ENVS[@]}" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
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
one c2_new $L/libafquant.so -- --workload configs2
one c2_s5 $L/libafquant_s5.so -- --workload configs2
one c2_s5fb6 $L/libafquant_s5fb6.so -- --workload configs2
one c2_s5fb12 $L/libafquant_s5fb12.so -- --workload configs2
one c2_s4fb12 $L/libafquant_s4fb12.so -- --workload configs2
one c2t_new $L/libafquant.so -- --workload configs2 --na-model tail
one c1_new $L/libafquant.so -- --steps 10 --warmup 3
one c1_sortcall $L/libafquant_sortcall.so -- --steps 10 --warmup 3
one c1t_new $L/libafquant.so -- --na-model tail
one c1t_sortcall $L/libafquant_sortcall.so -- --na-model tail
ls $O
