#!/bin/bash
# round 6, call 25: the staging threads sleep between pieces (condition variable)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_25; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_crlike.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -6 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
AFQ_HOST_TIMING=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also cli,e2e,cli_pug,configs2 > $O/line.json 2> $O/cli_timing.txt
grep "afquant\]" $O/cli_timing.txt | tail -12
python -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print({k: (v.get('wall_s'), v.get('value'), v.get('ms_per_step')) for k, v in d['also'].items() if isinstance(v, dict)})"
