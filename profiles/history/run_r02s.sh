#!/bin/bash
# fixed-slab capacity sweep on the default workload
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
for cap in 384; do
AFQ_SLAB_CAP=$cap timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also none > gpurun_out/r02s.json 2> gpurun_out/r02s.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02s.json') if l.startswith('{')][-1]);k=d['roofline']['all_kernels_ms_per_step'];print($cap, d['ms_per_step'], d['value'], k['k_scatter'], k['k_resolve'], k.get('k_fix_slabs'), d['overflow_buckets'])"
done
AFQ_FIXED_SLABS=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also none > gpurun_out/r02s.json 2> gpurun_out/r02s.err
python -c "
import json;d=json.loads([l for l in open('gpurun_out/r02s.json') if l.startswith('{')][-1]);k=d['roofline']['all_kernels_ms_per_step'];print('exact', d['ms_per_step'], d['value'], k['k_scatter'], k['k_resolve'], k['k_hist'])"
