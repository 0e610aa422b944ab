#!/bin/bash
# many-gene buckets through the two-table path: crlike tests, the tail bench with and without it, bucket targets
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 1200 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_multi.py tests/test_gpu_fuzz.py -q -x > gpurun_out/r04p_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04p_pytest.log
run() { env "$@" timeout 300 python bench.py --workload configs1 --na-model tail --steps 4 --also none --cpu-seconds 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', d['ms_per_step'], {k:v for k,v in d['roofline']['all_kernels_ms_per_step'].items() if v>0.2}, 'cpu ok' if d.get('cpu_baseline') else '')"; }
run AFQ_RESOLVE_NO_H2=1
run A=1
run AFQ_BUCKET_TARGET=192
run AFQ_BUCKET_TARGET=160
run AFQ_BUCKET_TARGET=128
