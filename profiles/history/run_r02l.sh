#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
( time timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_cli.py -m gpu -q -x ) > gpurun_out/r02l_pytest.log 2>&1
tail -5 gpurun_out/r02l_pytest.log | head -3
AFQ_HOST_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --also cli --steps 2 --warmup 1 2> gpurun_out/r02l.err | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print(d['also']['cli'])"
grep "afquant\]" gpurun_out/r02l.err | tail -7
