#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
bash profiles/run_timeline.sh r04h_c1
timeout 900 python -m pytest tests/test_gpu_multi.py -q -x -k "eight_ranks or spawns" > gpurun_out/r04h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04h_pytest.log
