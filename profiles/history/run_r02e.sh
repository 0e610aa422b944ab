#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q -x ) > gpurun_out/r02e_pytest.log 2>&1
tail -4 gpurun_out/r02e_pytest.log
timeout 600 python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --also none > gpurun_out/r02e_cfg2.json 2> gpurun_out/r02e_cfg2.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02e_cfg2.json') if l.startswith('{')][-1]);print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"
AFQ_LIB_PATH=$ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --workload configs2 --steps 1 --warmup 0 --no-cpu-baseline --also none 2>&1 | grep "pug " | cut -c1-330 | head -10
