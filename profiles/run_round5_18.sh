#!/bin/bash
# round 5, call 18: cover_tiny8 parks its short molecules until the batch's last round and resolves them a lane each
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_18; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -4 $O/tests.log | grep -v "^$"
( timeout 200 python tests/extended_fuzz.py 40 0; timeout 200 python tests/extended_fuzz.py 20 2000 ) > $O/fuzz.log 2>&1; grep "extended fuzz" $O/fuzz.log; grep -c FAILED $O/fuzz.log
L=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc
one() {  # name, extra env..., then bench flags after --
  local N=$1; shift
  local ENVS=()
  while [ "$1" != "--" ]; do ENVS+=("$1"); shift; done; shift
  env "${ENVS[@]}" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none "$@" > $O/$N.json 2> $O/$N.err
  python - "$N" "$O/$N.json" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = d["roofline"]["all_kernels_ms_per_step"]
    print(sys.argv[1], d["ms_per_step"], {a: round(b, 2) for a, b in k.items() if b > 0.3}, "mono", d["retries"].get("cells_through_the_one_workgroup_kernel"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
one c2 X=1 -- --workload configs2
one c2t X=1 -- --workload configs2 --na-model tail
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none --workload configs2 > /dev/null 2> $O/stats.err
cd $GRAFT_REPO_ROOT; python - <<'P'
import sqlite3, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/round5_18/stats/**/*results.db"), recursive=True)
if f:
    db = sqlite3.connect(f[0])
    for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if pct > 0.7: print(f"{name.split('(')[0][-40:]:42s} {calls:5d} {tot/1e3:10.1f} us {avg/1e3:9.1f} {pct:6.2f}")
P
find $O -size +8M -delete
