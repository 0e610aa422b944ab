#!/bin/bash
# range init as one copy + one kernel, copies on SDMA: crlike + multi tests, headline timeline, headline ms/step
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_em.py tests/test_gpu_cli.py -q -x > gpurun_out/r04j_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04j_pytest.log
bash profiles/run_timeline.sh r04j_c1 > /dev/null 2>&1
grep -E "^# " gpurun_out/tl_r04j_c1/timeline.txt | head -14
timeout 300 python bench.py --steps 20 --warmup 3 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
