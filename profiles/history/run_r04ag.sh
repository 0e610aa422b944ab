#!/bin/bash
# late round 4: measurement builds on the headline, steady state (20 steps after 5): wave -> slab-group mapping of the decoder
# (kDecodeCols), run of consecutive tiles per XCD in the scatter
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ag; mkdir -p $O
python -c "import torch" 2>/dev/null
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:22s} {d['ms_per_step']:7.3f} ms  ksum {sum(k.values()):.2f} " + " ".join(f"{a[2:]}={b:.3f}" for a, b in k.items() if b >= 0.02))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  run default_$rep AFQ_X=0
  for v in cols1 cols512 cols64k run4 run8 run32; do run ${v}_$rep AFQ_LIB_PATH=$L/libafquant_$v.so; done
done
