#!/bin/bash
# round 6, the evidence of the round's last kernel commit: the -m gpu suite, rocprofv3 stats / SQ / FETCH / WRITE passes of all six
# legs (-> profiles/r06_*), and the whole default bench line (-> profiles/r06_bench.json).  ~12 GPU-minutes.
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round6_final; mkdir -p $O
leg() {   # tag, bench flags...
  local N=$1; shift
  PASSES="stats sq fetch write" bash profiles/run_prof.sh $N "$@" > /dev/null 2>&1
  python profiles/summarize.py $N > $O/${N}_rocprof.txt 2>&1
  python profiles/traffic.py $N $N $O > /dev/null 2>&1
  cp gpurun_out/prof_$N/bench_stats.json $O/${N}_bench_under_rocprof.json 2>/dev/null
  rm -rf gpurun_out/prof_$N
  echo "$N: $(sed -n 3p $O/${N}_rocprof.txt | cut -c1-100)"
}
leg r06
leg r06_configs2 --workload configs2
leg r06_configs1_tail --na-model tail
leg r06_configs2_tail --workload configs2 --na-model tail
leg r06_configs3 --workload configs3
leg r06_atac --workload atac
cd "$GRAFT_REPO_ROOT"
cp $O/r06*_traffic.json $O/r06*_rocprof.txt profiles/ 2>/dev/null   # (the line's roofline.traffic and roofline.kernel are looked up in profiles/: this box's passes)
# (the six legs' passes first, then the line - it reads their traffic and kernel names -, the suite last)
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench.json 2> $O/bench.err; tail -c 1500 $O/r06_bench.json; echo; tail -3 $O/bench.err
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1; tail -4 $O/tests.log | grep -v "^$"
ls $O
