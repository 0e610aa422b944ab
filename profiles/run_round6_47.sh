#!/bin/bash
# round 6, call 47: staged host->device copies as independent lanes (a thread reads a piece into one of its two pinned pieces and queues the
# copy on its own stream): parity suites with the staged path forced onto small inputs (1 MiB pieces), then `afquant quant` on the 6.9 GB sample
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_47; mkdir -p $O
( AFQ_TEST_STAGE_PIECE_MB=1 AFQ_TEST_STAGE_THREADS=5 timeout 1500 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_cli.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py tests/test_gpu_atac.py -x -q -m gpu 2>&1 | tail -4 ) | tee $O/tests_forced.log
( timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 ) | tee $O/tests.log
for cfg in "0 8" "16 8" "8 8" "16 16" "16 4" "32 8" "12 32"; do
  set -- $cfg
  echo "== threads $1 (0: default) piece $2 MiB"
  AFQ_TEST_STAGE_THREADS=$1 AFQ_TEST_STAGE_PIECE_MB=$2 AFQ_HOST_TIMING=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also cli 2> $O/err_$1_$2.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['legs']['cli'])"
  grep -i "device batches\|busy" $O/err_$1_$2.txt | tail -2
done 2>&1 | tee $O/cli.txt
