#!/bin/bash
# lab note: FETCH_SIZE / WRITE_SIZE passes of one leg only (default configs2), per-kernel GB per step printed; $1 = tag, rest = bench flags
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-x}; shift
O=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG; mkdir -p $O
PASSES="fetch write" bash profiles/run_prof.sh t_$TAG "$@" > /dev/null 2>&1
python profiles/traffic.py t_$TAG t_$TAG $O > /dev/null 2>&1
rm -rf gpurun_out/prof_t_$TAG
python - <<PY
import json,glob
d=json.load(open(glob.glob("$O/*traffic.json")[0]))
steps=d['steps_profiled']
rows=sorted(((v['bytes_total_fetch_doubled']/steps/1e9,k,v['FETCH_SIZE_KB_per_launch']*2048*v['dispatches']/steps/1e9,v['WRITE_SIZE_KB_per_launch']*1024*v['dispatches']/steps/1e9) for k,v in d['kernels'].items()),reverse=True)
for r in rows[:14]: print("%8.2f GB/step  %-26s fetch(x2) %.2f write %.2f"%r)
print("total %.1f"%sum(r[0] for r in rows))
PY
