#!/bin/bash
# late round 4: k_p2_lone with labels of 5..64 refs resolved by the wave (AFQ_P2_LONE_COOP) - the parsimony tests, the tailed
# fuzz family (every decoder / dedup / lone setting), configs2_tail both ways; the headline through the lane-per-dword decoder
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ai; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_pug.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 300 python tests/extended_fuzz.py 42 2000 > $O/fuzz.log 2>&1; tail -6 $O/fuzz.log
export AFQ_BENCH_CRC=1
run() {  # name, flags, env...
  local name=$1 flags=$2; shift 2
  env "$@" timeout 200 python bench.py $flags --also none --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:14s} {d['ms_per_step']:7.3f} ms  crc {d.get('rows_crc32')}  ksum {sum(k.values()):.2f} " + " ".join(f"{a[2:]}={b:.3f}" for a, b in k.items() if b >= 0.02))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
T="--workload configs2 --na-model tail --steps 2 --warmup 1"
run c2tail_coop "$T" AFQ_X=0
run c2tail_lane "$T" AFQ_P2_LONE_COOP=0
run c2_coop "--workload configs2 --steps 2 --warmup 1" AFQ_X=0
run head_keyshash "--steps 10 --warmup 3" AFQ_DECODE=keys
run head_recs "--steps 10 --warmup 3" AFQ_X=0
