#!/bin/bash
# late round 4: k_decode_keys with the hash-table dedup (AFQ_DECODE_DEDUP=hash) - its tests, then the label-tail workload (configs1_tail) both ways
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ah; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_crlike.py -m gpu -q -k "dedups or straddling" > $O/tests.log 2>&1; tail -5 $O/tests.log
export AFQ_BENCH_CRC=1
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --na-model tail --also none --no-cpu-baseline --steps 6 --warmup 2 > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:12s} {d['ms_per_step']:7.3f} ms  crc {d.get('rows_crc32')}  ksum {sum(k.values()):.2f} " + " ".join(f"{a[2:]}={b:.3f}" for a, b in k.items() if b >= 0.02))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  run scan_$rep AFQ_X=0
  run hash_$rep AFQ_DECODE_DEDUP=hash
done
