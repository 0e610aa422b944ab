#!/bin/bash
# late round 4, second pass (the next range no longer starts beside the histograms): the range pipeline's switches one by one on the headline (ms per step, kernel brackets, the rows' CRC-32),
# the host laps of one step, configs2 with and without them; the -m gpu suite first
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ab; mkdir -p $O
python -c "import torch" 2>/dev/null
( time timeout 420 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1
tail -15 $O/tests.log | grep -v "^$"
export AFQ_BENCH_CRC=1
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --also none --no-cpu-baseline --steps 10 --warmup 3 > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:28s} {d['ms_per_step']:7.3f} ms  crc {d.get('rows_crc32')}  frac {d['roofline']['frac']:.4f} ksum {sum(k.values()):.2f} " + " ".join(f"{a[2:]}={b:.2f}" for a, b in k.items() if b >= 0.02))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run default AFQ_X=0
run pair AFQ_TIMER_MODE=pair
run chain AFQ_TIMER_MODE=chain
run no_chain_compact AFQ_CHAIN_COMPACT=0
run host_tables AFQ_DEVICE_TABLES=0
run all_old AFQ_TIMER_MODE=pair AFQ_CHAIN_COMPACT=0 AFQ_DEVICE_TABLES=0
run no_init_sync AFQ_INIT_SYNC=0
run taper6 AFQ_CR_TAPER=0.26,0.52,0.74,0.88,0.96
run taper6b AFQ_CR_TAPER=0.24,0.48,0.70,0.86,0.95
run taper7 AFQ_CR_TAPER=0.22,0.44,0.64,0.80,0.91,0.97
run taper5s AFQ_CR_TAPER=0.30,0.58,0.80,0.94
run default_again AFQ_X=0
AFQ_HOST_TIMING=1 timeout 120 python bench.py --also none --no-cpu-baseline --steps 1 --warmup 2 > $O/laps.json 2> $O/laps.err
awk '/submit: chunk headers/{n++} n>=3' $O/laps.err | head -40
# configs2: new default against the old arrangement
for v in new old; do
  if [ $v = old ]; then export AFQ_TIMER_MODE=pair AFQ_CHAIN_COMPACT=0 AFQ_DEVICE_TABLES=0; fi
  timeout 200 python bench.py --workload configs2 --also none --no-cpu-baseline --steps 3 --warmup 1 > $O/c2_$v.json 2> $O/c2_$v.err
  python - $v $O/c2_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("configs2", sys.argv[1], d["ms_per_step"], "crc", d.get("rows_crc32"))
except Exception as e:
    print("configs2", sys.argv[1], "FAILED", e)
PY
done
