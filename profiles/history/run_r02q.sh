#!/bin/bash
# cr-like pipeline check: parity tests + the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -q -x ) > gpurun_out/r02q_pytest.log 2>&1
tail -4 gpurun_out/r02q_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also none > gpurun_out/r02q_bench.json 2> gpurun_out/r02q_bench.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02q_bench.json') if l.startswith('{')][-1]);print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"
