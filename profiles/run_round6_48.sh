#!/bin/bash
# round 6, call 48: the staged copy as it was (fillers meet per piece) with sixteen fillers and 32 MiB pieces: front-end suites, the staged path
# forced onto small inputs, `afquant quant` on the 6.9 GB sample three times
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_48; mkdir -p $O
( AFQ_TEST_STAGE_PIECE_MB=1 AFQ_TEST_STAGE_THREADS=5 timeout 1500 python -m pytest tests/test_gpu_crlike.py tests/test_gpu_cli.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3 ) | tee $O/tests_forced.log
( timeout 1500 python -m pytest tests/test_gpu_cli.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 ) | tee $O/tests.log
for i in 1 2 3; do
  AFQ_HOST_TIMING=1 timeout 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --also cli,cli_pug 2> $O/err_$i.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['legs']['cli'], d['legs'].get('cli_pug'))"
done 2>&1 | tee $O/cli.txt
