#!/bin/bash
# pack written straight to pinned memory (no copy in the queue in front of the next range's upload); taper variants; timeline
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_em.py -q -x > gpurun_out/r04v_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04v_pytest.log
run() { env "$@" timeout 300 python bench.py --steps 25 --warmup 3 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', d['ms_per_step'])"; }
run A=1
run AFQ_CR_TAPER=0.28,0.56,0.78,0.92,0.97
run AFQ_CR_TAPER=0.30,0.58,0.80,0.93,0.98
run AFQ_CR_TAPER=0.25,0.50,0.72,0.88,0.96,0.99
run A=2
bash profiles/run_timeline.sh r04v_c1 > /dev/null 2>&1
grep -E "^# " gpurun_out/tl_r04v_c1/timeline.txt | head -4
