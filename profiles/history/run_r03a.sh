#!/bin/bash
# round 3, first contact of the phase-kernel parsimony path with the GPU: the pug tests on all three routes, then configs2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_gpu_pug.py -x -q 2>&1 | tail -25 > gpurun_out/r03a_pytest.log
cat gpurun_out/r03a_pytest.log
timeout 300 python bench.py --workload configs2 --steps 2 --warmup 1 --also none > gpurun_out/r03a_cfg2.json 2> gpurun_out/r03a_cfg2.err
tail -c 3000 gpurun_out/r03a_cfg2.json; tail -5 gpurun_out/r03a_cfg2.err
