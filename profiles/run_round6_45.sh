#!/bin/bash
# round 6, call 45-46: `afquant quant` on the 6.9 GB sample - the staging of the input (pread into pinned pieces by N threads, a DMA per
# piece): N threads x piece size, AFQ_HOST_TIMING=1
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_45; mkdir -p $O
for cfg in "8 16" "8 8" "16 8" "4 16" "8 32" "16 32" "16 4" "12 16"; do
  set -- $cfg
  echo "== threads $1 piece $2 MiB"
  AFQ_TEST_STAGE_THREADS=$1 AFQ_TEST_STAGE_PIECE_MB=$2 AFQ_HOST_TIMING=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also cli 2> $O/err_$1_$2.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['legs']['cli'])"
  grep -i "device\|submit\|batch" $O/err_$1_$2.txt | tail -4
done 2>&1 | tee $O/cli.txt
