#!/bin/bash
# round 2: the whole -m gpu suite, the default bench line, and the committed rocprof evidence (default workload + configs[2]).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02z_pytest.log 2>&1
tail -4 gpurun_out/r02z_pytest.log | head -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > gpurun_out/r02z_smoke.log 2>&1; tail -1 gpurun_out/r02z_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r02z_bench.json 2> gpurun_out/r02z_bench.err
PASSES="stats sq sq2 fetch write" bash profiles/run_prof.sh r2z --workload configs1 > gpurun_out/r02z_prof.log 2>&1
PASSES="stats sq sq2 fetch write" bash profiles/run_prof.sh r2z_cfg2 --workload configs2 > gpurun_out/r02z_prof_cfg2.log 2>&1
timeout 600 python bench.py --workload atac --steps 3 --warmup 1 > gpurun_out/r02z_bench_atac.json 2> gpurun_out/r02z_bench_atac.err
timeout 600 python bench.py --gpus 2 --share-gpu --dist-backend gloo --steps 2 --warmup 1 --cells 2000 --c3-cells 20000 --also configs3 > gpurun_out/r02z_bench_2ranks_shared.json 2> gpurun_out/r02z_bench_2ranks_shared.err
tail -c 300 gpurun_out/r02z_bench_atac.json
AFQ_LIB_PATH=$ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --workload configs2 --steps 1 --warmup 0 --no-cpu-baseline --also none 2>&1 | grep -E "^pug |^em " | cut -c1-420 | sort | uniq > gpurun_out/r02z_phase_clocks.txt
