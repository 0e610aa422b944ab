#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
( time timeout 1500 python -m pytest tests/test_gpu_atac.py tests/test_gpu_crlike.py -m gpu -q -x ) > gpurun_out/r02n_pytest.log 2>&1
head -3 gpurun_out/r02n_pytest.log
timeout 600 python bench.py --workload atac --steps 3 --warmup 1 > gpurun_out/r02n_bench_atac.json 2> gpurun_out/r02n_bench_atac.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02n_bench_atac.json') if l.startswith('{')][-1]);print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'])"
