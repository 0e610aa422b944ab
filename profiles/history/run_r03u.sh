#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03u; mkdir -p $O
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
for i in 1 2; do timeout 120 python bench.py --workload configs2 $B > $O/cfg2_$i.json 2> $O/cfg2.err; show $O/cfg2_$i.json; done
AFQ_P2_GRID=4096 timeout 120 python bench.py --workload configs2 $B > $O/cfg2_g4096.json 2> $O/cfg2.err; show $O/cfg2_g4096.json
AFQ_P2_GRID=16384 timeout 120 python bench.py --workload configs2 $B > $O/cfg2_g16384.json 2> $O/cfg2.err; show $O/cfg2_g16384.json
AFQ_LIB_PATH=$PWD/alevin-fry_amd/csrc/libafquant_timing.so timeout 120 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline 2>/dev/null | grep -E "^p2 graph|^em " | head -60 > $O/clocks.txt
grep "^p2 graph" $O/clocks.txt | sed -n '1,4p;12,16p'
