#!/bin/bash
# which engine copies the CSR back: blit kernels (on the compute queue, next to the next range's decode) or SDMA?  env experiments
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
for V in default blit0 ; do
  case $V in
    default) E="" ;;
    blit0) E="GPU_FORCE_BLIT_COPY_SIZE=0" ;;
  esac
  echo "== $V"
  env $E bash profiles/run_timeline.sh r04i_$V > /dev/null 2>&1
  grep -E "^# (K __amd|C MEMORY|covered|last)" gpurun_out/tl_r04i_$V/timeline.txt
  env $E timeout 300 python bench.py --steps 10 --warmup 2 --also none --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
done
