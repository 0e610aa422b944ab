#!/bin/bash
# full -m gpu suite + the configs[2] line (after the wide-class emission)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02y_pytest.log 2>&1
tail -4 gpurun_out/r02y_pytest.log | head -2
timeout 600 python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --also none > gpurun_out/r02y_cfg2.json 2> gpurun_out/r02y_cfg2.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02y_cfg2.json') if l.startswith('{')][-1]);print(d['ms_per_step'], d['value'], d['roofline']['all_kernels_ms_per_step'])"
