#!/bin/bash
# round 5, call 8: scATAC in six growing ranges; a dozen workgroups of k_pug_cell until a cell reaches it; k_p2_search compiled for seven workgroups per CU (variant)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_08; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_atac.py tests/test_gpu_pug.py tests/test_gpu_cli.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"
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
one c2 $L/libafquant.so -- --workload configs2
one c2_s7 $L/libafquant_s7.so -- --workload configs2
one c2t_s7 $L/libafquant_s7.so -- --workload configs2 --na-model tail
timeout 300 python bench.py --workload atac --steps 3 --warmup 1 --no-cpu-baseline > $O/atac.json 2> $O/atac.err; python -c "
import json; d=json.load(open('$O/atac.json')); print('atac', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
find $O -size +8M -delete
