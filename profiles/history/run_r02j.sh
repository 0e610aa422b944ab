#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time AFQ_HOST_TIMING=1 timeout 900 python bench.py ) > gpurun_out/r02j_bench.json 2> gpurun_out/r02j_bench.err
grep "afquant\]" gpurun_out/r02j_bench.err | tail -12
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02j_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['also'].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','wall_s','h2d_GBps_effective','error','skipped')})
print(d['also']['configs2'].get('cpu_baseline'))
PY
