#!/bin/bash
# EM classes laid out longest label first + four label words per step: EM tests, tail + plain configs2
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_crlike.py -q -x -k "em or EM" > gpurun_out/r04s_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04s_pytest.log
for M in plain tail; do timeout 400 python bench.py --workload configs2 --na-model $M --steps 3 --also none --cpu-seconds 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M', 'ms_per_step', d['ms_per_step'], {k:v for k,v in d['roofline']['all_kernels_ms_per_step'].items() if v>1}, (d.get('cpu_baseline') or {}).get('em_arithmetic'))"; done
