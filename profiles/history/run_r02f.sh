#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_crlike.py -m gpu -q -x ) > gpurun_out/r02f_pytest.log 2>&1
tail -4 gpurun_out/r02f_pytest.log | head -2
for i in 1; do
timeout 600 python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --also none > gpurun_out/r02f_cfg2.json 2> gpurun_out/r02f_cfg2.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02f_cfg2.json') if l.startswith('{')][-1]);print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"
done
AFQ_NO_PIPELINE=1 timeout 600 python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --also none 2>/dev/null | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print('one range:', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
