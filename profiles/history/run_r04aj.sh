#!/bin/bash
# round 4, final evidence of the late-round code: the -m gpu suite, rocprofv3 stats / SQ / FETCH / WRITE passes of the two legs
# whose kernels changed (headline, configs1_tail), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r04aj; mkdir -p $O
( time timeout 420 python -m pytest tests -m gpu -q ) > $O/tests.log 2>&1; tail -6 $O/tests.log | grep -v "^$"
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
leg configs1_tail --na-model tail
cd "$GRAFT_REPO_ROOT"
( time timeout 500 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(name, x):
    r = x.get("roofline") or {}
    print(f"{name:16s} {x.get('value')} {x.get('unit')}  {x.get('ms_per_step')} ms  frac {r.get('frac')} ({r.get('kernel')}) traffic {r.get('traffic')} slowdown {x.get('slowdown_per_input_byte_vs_plain')}")
show("headline", d)
for k, v in d["also"].items():
    if isinstance(v, dict) and "ms_per_step" in v: show(k, v)
    else: print(k, json.dumps(v)[:200])
PY
ls $O
