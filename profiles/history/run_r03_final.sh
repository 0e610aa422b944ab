#!/bin/bash
# round 3, end-of-round run: the whole -m gpu suite, a short stress of configs[2], the default bench line with all its legs,
# rocprofv3 passes (stats / SQ / FETCH / WRITE) for configs[1] and configs[2], configs[2] with one range (kernels alone),
# the per-cell phase clocks of the instrumented build.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_final; mkdir -p $O
python -c "import torch" 2>/dev/null
timeout 500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash profiles/run_stress.sh 4 2>&1 | tail -4
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.json; echo
PASSES="stats fetch write sq" bash profiles/run_prof.sh r03f > $O/prof1.log 2>&1
PASSES="stats fetch write sq" bash profiles/run_prof.sh r03f_cfg2 --workload configs2 > $O/prof2.log 2>&1
mkdir -p gpurun_out/prof_r03f_cfg2_alone; cd /tmp && export TMPDIR=/tmp
AFQ_NO_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03f_cfg2_alone/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --workload configs2 --steps 3 --warmup 1 --no-cpu-baseline --also none > $GRAFT_REPO_ROOT/gpurun_out/prof_r03f_cfg2_alone/bench_stats.json 2> $GRAFT_REPO_ROOT/$O/alone.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_r03f_cfg2_alone -size +16M -delete
AFQ_LIB_PATH=$PWD/alevin-fry_amd/csrc/libafquant_timing.so timeout 120 python bench.py --workload configs2 --steps 1 --warmup 0 --also none --no-cpu-baseline 2>/dev/null | grep -E "^p2 graph|^em " | head -80 > $O/clocks.txt
wc -l $O/clocks.txt; du -sh gpurun_out/prof_r03f* | tail -3
