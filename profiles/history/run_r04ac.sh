#!/bin/bash
# late round 4: measurement builds on the headline (non-temporal loads of once-read data in decode / scatter / resolve, 4096-key
# scatter tiles, 2 / 8 slabs per decoder wave), three rounds, every build in every round (a step varies by +-0.4 ms between runs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ac; mkdir -p $O
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
for rep in 1 2 3; do
  run default_$rep AFQ_X=0
  for v in nt_dec nt_sc nt_res tile4k spw8 spw2; do run ${v}_$rep AFQ_LIB_PATH=$L/libafquant_$v.so; done
  run no_init_sync_$rep AFQ_INIT_SYNC=0
done
