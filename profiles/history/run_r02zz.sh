#!/bin/bash
# round 2, last refresh after the sort-primitive / resolve / EM-width changes: the whole -m gpu suite, smoke, the default
# bench line, rocprof kernel stats + HBM traffic counters for the default workload and configs[2], the ATAC line and
# the per-phase device clocks.  (The SQ instruction-mix counters in r02_rocprof.txt stay those of run_r02z.sh.)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r02zz_pytest.log 2>&1
tail -4 gpurun_out/r02zz_pytest.log | head -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > gpurun_out/r02zz_smoke.log 2>&1; tail -1 gpurun_out/r02zz_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r02zz_bench.json 2> gpurun_out/r02zz_bench.err
PASSES="stats fetch write" bash profiles/run_prof.sh r2zz --workload configs1 > gpurun_out/r02zz_prof.log 2>&1
PASSES="stats fetch write" bash profiles/run_prof.sh r2zz_cfg2 --workload configs2 > gpurun_out/r02zz_prof_cfg2.log 2>&1
timeout 600 python bench.py --workload atac --steps 3 --warmup 1 > gpurun_out/r02zz_bench_atac.json 2> gpurun_out/r02zz_bench_atac.err
tail -c 300 gpurun_out/r02zz_bench_atac.json
AFQ_LIB_PATH=$ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 600 python bench.py --workload configs2 --steps 1 --warmup 0 --no-cpu-baseline --also none 2>&1 | grep -E "^pug |^em " | cut -c1-420 | sort | uniq > gpurun_out/r02zz_phase_clocks.txt
tail -c 600 gpurun_out/r02zz_bench.json
