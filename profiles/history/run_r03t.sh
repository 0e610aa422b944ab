#!/bin/bash
# round 3: search v2 (one probe site, shared queue), lone-vertex gather chain, class minima without the hashed-vertex pass,
# k_cell_hist with one scan per pass; tail: bucket target / slab capacity combinations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03t; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
B="--steps 3 --warmup 1 --also none --no-cpu-baseline"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], {x:round(k[x],2) for x in k if k[x]>0.3})
except Exception as e: print(sys.argv[1], "fail", e)
PY
}
timeout 120 python bench.py --workload configs2 $B > $O/cfg2_v2.json 2> $O/cfg2_v2.err; show $O/cfg2_v2.json
AFQ_P2_SEARCH=v1 timeout 120 python bench.py --workload configs2 $B > $O/cfg2_v1.json 2> $O/cfg2_v1.err; show $O/cfg2_v1.json
timeout 120 python bench.py --steps 10 --warmup 3 --also none --no-cpu-baseline > $O/cfg1.json 2> $O/cfg1.err; show $O/cfg1.json
timeout 120 python bench.py --na-model tail $B > $O/tail_256_384.json 2> $O/tail_a.err; show $O/tail_256_384.json
AFQ_BUCKET_TARGET=192 timeout 120 python bench.py --na-model tail $B > $O/tail_192_384.json 2> $O/tail_b.err; show $O/tail_192_384.json
AFQ_BUCKET_TARGET=128 AFQ_SLAB_CAP=256 timeout 120 python bench.py --na-model tail $B > $O/tail_128_256.json 2> $O/tail_c.err; show $O/tail_128_256.json
AFQ_BUCKET_TARGET=128 AFQ_SLAB_CAP=192 timeout 120 python bench.py --na-model tail $B > $O/tail_128_192.json 2> $O/tail_d.err; show $O/tail_128_192.json
AFQ_LIB_PATH=$PWD/alevin-fry_amd/csrc/libafquant_timing.so timeout 120 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline 2>/dev/null | grep -E "^p2 graph|^em " | head -60 > $O/clocks.txt
wc -l $O/clocks.txt
