#!/bin/bash
# round 5, call 2: the hash-dedup partition kernel + union-find components against the round-4 library (libafquant_base.so),
# the search filter at 2^13 / 2^14 bits, the graph kernel's phase clocks on the plain model, the new EM-at-size tests.
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not configs3" -s ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"; grep "vs the reference arithmetic" $O/tests.log
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
one c2_base $L/libafquant_base.so -- --workload configs2
one c2_new $L/libafquant.so -- --workload configs2
one c2_new_sortpart $L/libafquant.so AFQ_P2_PART=sort -- --workload configs2
one c2_f13 $L/libafquant_f13.so -- --workload configs2
one c2_f14 $L/libafquant_f14.so -- --workload configs2
one c2t_base $L/libafquant_base.so -- --workload configs2 --na-model tail
one c2t_new $L/libafquant.so -- --workload configs2 --na-model tail
AFQ_LIB_PATH=$L/libafquant_timing.so timeout 300 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline > $O/plain_clocks.txt 2> $O/plain_clocks.err
grep "p2 graph" $O/plain_clocks.txt | sort -t= -k2 -n -r | awk 'NR%4==1' | head -14
ls $O
