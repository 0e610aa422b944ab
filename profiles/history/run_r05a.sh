#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py -q -x > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r05a_pytest.log
for i in 1 2; do timeout 300 python bench.py --workload configs2 --steps 4 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'graph', d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"; done
timeout 300 python bench.py --workload configs2 --na-model tail --steps 2 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail ms_per_step', d['ms_per_step'], 'graph', d['roofline']['all_kernels_ms_per_step']['k_p2_graph'])"
