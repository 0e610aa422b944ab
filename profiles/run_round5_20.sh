#!/bin/bash
# round 5, call 20: the -m gpu suite and smoke() on the round's last commit (after the evidence run only the planner's EM budget and a build macro changed)
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_20; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1; tail -4 $O/tests.log | grep -v "^$"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -3
timeout 300 python bench.py --gpus 2 --share-gpu --dist-backend gloo --plan-only --also none 2>/dev/null | cut -c1-400
