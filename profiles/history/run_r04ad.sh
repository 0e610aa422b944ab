#!/bin/bash
# late round 4: histograms on a high-priority stream beside the next range's decoder (AFQ_TAIL_OVERLAP=2), geometric tapers,
# 16 slabs per decoder wave, the two measurement builds that won (8 slabs per wave + non-temporal key loads in the resolve) together
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ad; mkdir -p $O
python -c "import torch" 2>/dev/null
export AFQ_BENCH_CRC=1
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 3 > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:22s} {d['ms_per_step']:7.3f} ms  crc {d.get('rows_crc32')}  ksum {sum(k.values()):.2f} " + " ".join(f"{a[2:]}={b:.3f}" for a, b in k.items() if b >= 0.02))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  run default_$rep AFQ_X=0
  run tail2_$rep AFQ_TAIL_OVERLAP=2
  run taper6g_$rep AFQ_CR_TAPER=0.419,0.671,0.822,0.913,0.967
  run taper7g_$rep AFQ_CR_TAPER=0.368,0.607,0.763,0.864,0.930,0.972
  run taper8g_$rep AFQ_CR_TAPER=0.318,0.540,0.696,0.805,0.881,0.935,0.972
  for v in spw16 combo combo16; do run ${v}_$rep AFQ_LIB_PATH=$L/libafquant_$v.so; done
  run combo_tail2_$rep AFQ_LIB_PATH=$L/libafquant_combo.so AFQ_TAIL_OVERLAP=2
done
