#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_pug.py -q -x -k "collision or outgrows or narrow or skew" > gpurun_out/r05c_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05c_pytest.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_cli.py -q -x > gpurun_out/r05c_pytest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05c_pytest2.log
