#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for T in 8 32 64; do
  echo "== AFQ_STAGE_THREADS=$T"
  AFQ_STAGE_THREADS=$T AFQ_HOST_TIMING=1 timeout 600 python bench.py --no-cpu-baseline --also cli --steps 2 --warmup 1 2> gpurun_out/r02k_$T.err | python -c "
import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);print(d['also']['cli'])"
  grep "afq_submit" gpurun_out/r02k_$T.err | tail -1
done
