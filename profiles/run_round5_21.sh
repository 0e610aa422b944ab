#!/bin/bash
# round 5, call 21: scATAC without the ref column on the link (run list + host fill), 6 / 7 / 8 growing ranges; the EM's streamed class pass at 4 / 6 / 8 classes per thread and trip
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_21; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_atac.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -5 $O/tests.log | grep -v "^$"
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
atac() {
  env AFQ_LIB_PATH=$2 timeout 300 python bench.py --workload atac --steps 5 --warmup 2 --no-cpu-baseline > $O/$1.json 2> $O/$1.err
  python -c "
import json; d=json.load(open('$O/$1.json')); print('$1', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])" 2>&1 | tail -1
}
atac atac $L/libafquant.so
atac atac_g1 $L/libafquant_g1.so
atac atac_g2 $L/libafquant_g2.so
atac atac_again $L/libafquant.so
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
for v in "" _em6 _em8; do
  one c2$v $L/libafquant$v.so --workload configs2
  one c2t$v $L/libafquant$v.so --workload configs2 --na-model tail
done
find $O -size +8M -delete
