#!/bin/bash
# round 6, call 23: where `afquant quant` spends its second (AFQ_HOST_TIMING=1 on the cli leg)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_23; mkdir -p $O
AFQ_HOST_TIMING=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also cli > $O/line.json 2> $O/cli_timing.txt
grep -v "^\[bench\]" $O/cli_timing.txt | tail -80
python -c "
import json; d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]); print(d['also']['cli'])"
