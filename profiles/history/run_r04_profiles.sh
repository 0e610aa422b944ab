#!/bin/bash
# Round 4 evidence: per leg of the default bench line - rocprofv3 --kernel-trace --stats, an SQ pass, FETCH_SIZE and WRITE_SIZE passes
# (each pass its own run, counters never combined with other traces), summarised on the box into gpurun_out/r04_profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r04_profiles; mkdir -p $O
leg() {   # name, bench flags...
  local N=$1; shift
  PASSES="stats sq fetch write" bash profiles/run_prof.sh r04_$N "$@" > /dev/null 2>&1
  python profiles/summarize.py r04_$N > $O/r04_${N}_rocprof.txt 2>&1
  python profiles/traffic.py r04_$N r04_$N $O > /dev/null 2>&1
  cp gpurun_out/prof_r04_$N/bench_stats.json $O/r04_${N}_bench_under_rocprof.json 2>/dev/null
  rm -rf gpurun_out/prof_r04_$N
  echo "$N done: $(head -3 $O/r04_${N}_rocprof.txt | tail -1)"
}
leg configs1
leg configs2 --workload configs2
leg configs1_tail --na-model tail
leg configs2_tail --workload configs2 --na-model tail
leg configs3 --workload configs3
leg atac --workload atac
ls -la $O
