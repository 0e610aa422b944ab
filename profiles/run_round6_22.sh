#!/bin/bash
# round 6, call 22: the scATAC sort in registers (k_atac_dedup64: atac_reg_net)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_22; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_atac.py tests/test_gpu_prims.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
PASSES="stats" bash profiles/run_prof.sh r6s_atac --workload atac > /dev/null 2>&1
python profiles/summarize.py r6s_atac > $O/r6s_atac_rocprof.txt 2>&1
head -12 $O/r6s_atac_rocprof.txt | cut -c1-100; tail -3 $O/r6s_atac_rocprof.txt | cut -c1-400
rm -rf gpurun_out/prof_r6s_atac
timeout 600 python bench.py --steps 3 --warmup 1 --also none --workload atac 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('atac', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d.get('cpu_baseline',{}).get('value'))"
