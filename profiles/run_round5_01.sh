#!/bin/bash
# Prepared at the end of round 4 for the FIRST GPU call of the next round (a call costs one to two GPU-minutes: ~4 minutes for this one).
# Needs `make -C alevin-fry_amd/csrc timing` on the build host first (libafquant_timing.so travels with the snapshot).
#  1. the -m gpu suite on the round's starting commit;
#  2. rocprofv3 stats / SQ / FETCH / WRITE passes of the legs whose committed passes predate the late round-4 changes
#     (configs2, configs2_tail: k_p2_lone<true>, TimerChain; configs3: six geometric ranges) -> profiles/r05_*;
#  3. device phase clocks of the graph / cover kernels and of the EM on the tailed model (what the 110 + 102 ms are made of);
#  4. every step's own time of the default line (the early 7 ms hiccup, run_r04af.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r05_first; mkdir -p $O
( time timeout 420 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1; tail -6 $O/tests.log | grep -v "^$"
leg() {   # name, bench flags...
  local N=$1; shift
  PASSES="stats sq fetch write" bash profiles/run_prof.sh r05_$N "$@" > /dev/null 2>&1
  python profiles/summarize.py r05_$N > $O/r05_${N}_rocprof.txt 2>&1
  python profiles/traffic.py r05_$N r05_$N $O > /dev/null 2>&1
  cp gpurun_out/prof_r05_$N/bench_stats.json $O/r05_${N}_bench_under_rocprof.json 2>/dev/null
  rm -rf gpurun_out/prof_r05_$N
  echo "$N done: $(head -3 $O/r05_${N}_rocprof.txt | tail -1)"
}
leg configs2 --workload configs2
leg configs2_tail --workload configs2 --na-model tail
# configs3 is retaken with all legs at the end of the round (profiles/run_r05_final.sh)
cd "$GRAFT_REPO_ROOT"
if [ -f alevin-fry_amd/csrc/libafquant_timing.so ]; then
  AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --na-model tail --steps 1 --warmup 0 --also none --no-cpu-baseline > $O/tail_clocks.txt 2> $O/tail_clocks.err
  grep -c "p2 graph" $O/tail_clocks.txt; grep "p2 graph" $O/tail_clocks.txt | sort -t= -k2 -n -r | head -12
  grep "em2" $O/tail_clocks.txt | sort -t= -k2 -n -r | head -12
fi
AFQ_BENCH_STEP_TIMES=1 timeout 120 python bench.py --also none --no-cpu-baseline --steps 8 --warmup 0 2>&1 >/dev/null | grep "\[bench\] step" | awk '{printf "%s ", $3} END {print ""}'
ls $O
