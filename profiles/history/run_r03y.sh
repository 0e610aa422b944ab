#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03y; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_pug.py tests/test_gpu_fullsize.py -m gpu -x -q  > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
B="--steps 3 --warmup 1 --also none --no-cpu-baseline --workload configs2"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], {x:round(k[x],1) for x in k if k[x]>3})
except Exception as e: print(sys.argv[1], "fail", e)
PY
}
run() { name=$1; shift; env "$@" timeout 120 python bench.py $B > $O/$name.json 2> $O/$name.err; show $O/$name.json; }
run base A=1
run base2 A=1
run overlap AFQ_RANGE_OVERLAP=1
AFQ_LIB_PATH=$PWD/alevin-fry_amd/csrc/libafquant_timing.so timeout 120 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline 2>/dev/null | grep -E "^em " | head -40 > $O/clocks.txt
grep "^em rounds" $O/clocks.txt | head -9
