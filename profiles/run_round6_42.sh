#!/bin/bash
# round 6, call 42: the new parity test of k_p2_check's clearing (signature near-misses), through the seven routes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_42; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_pug.py -q -m gpu -k "share_no_ref" 2>&1 | tail -8 ) | tee $O/tests.log
