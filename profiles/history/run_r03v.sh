#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03v; mkdir -p $O
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
for i in 1 2 3; do timeout 120 python bench.py --workload configs2 $B > $O/cfg2_$i.json 2> $O/cfg2.err; show $O/cfg2_$i.json; done
AFQ_LIB_PATH=$PWD/alevin-fry_amd/csrc/libafquant_timing.so timeout 120 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline 2>/dev/null | grep -E "^p2 graph|^em " | head -70 > $O/clocks.txt
grep "^p2 graph" $O/clocks.txt | sed -n '1,2p;12,18p'; grep "^em rounds" $O/clocks.txt | head -8
