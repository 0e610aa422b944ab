#!/bin/bash
# k_p2_search with branch-free filter checks: pug tests, configs2 kernel times (two runs)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x > gpurun_out/r04w_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04w_pytest.log
for i in 1 2; do timeout 300 python bench.py --workload configs2 --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], {k:v for k,v in d['roofline']['all_kernels_ms_per_step'].items() if v>1})"; done
timeout 300 python bench.py --workload configs2 --na-model tail --steps 2 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail ms_per_step', d['ms_per_step'], {k:v for k,v in d['roofline']['all_kernels_ms_per_step'].items() if v>1})"
