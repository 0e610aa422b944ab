#!/bin/bash
# the whole -m gpu suite + headline timeline and ms/step after the range-init / pack-small changes
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04k_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04k_pytest.log
bash profiles/run_timeline.sh r04k_c1 > /dev/null 2>&1
grep -E "^# " gpurun_out/tl_r04k_c1/timeline.txt | head -12
timeout 300 python bench.py --steps 20 --warmup 3 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
