#!/bin/bash
# round 4, last GPU call: k_p2_lone with labels of 5..8 refs resolved by their own lane in registers (AFQ_P2_LONE_COOP=2):
# twelve tailed parsimony workloads against the oracle, then configs2_tail and configs2 with it and without
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04ao; mkdir -p $O
timeout 60 python tests/extended_fuzz.py 12 3000 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
export AFQ_BENCH_CRC=1
run() {  # name, flags, env...
  local name=$1 flags=$2; shift 2
  env "$@" timeout 60 python bench.py $flags --also none --no-cpu-baseline > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(f"{sys.argv[1]:12s} {d['ms_per_step']:7.3f} ms  crc {d.get('rows_crc32')}  lone={k.get('k_p2_lone')}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run tail_l8 "--workload configs2 --na-model tail --steps 2 --warmup 1" AFQ_P2_LONE_COOP=2
run tail_coop "--workload configs2 --na-model tail --steps 2 --warmup 1" AFQ_X=0
run plain_l8 "--workload configs2 --steps 2 --warmup 1" AFQ_P2_LONE_COOP=2
