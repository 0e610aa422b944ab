#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/prof_r2g
AFQ_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r2g/stats -o stats -- python $ROOT/bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --also none > $ROOT/gpurun_out/prof_r2g/bench.json 2> $ROOT/gpurun_out/prof_r2g/err.txt
cd $ROOT
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/prof_r2g/stats/*results.db')[0])
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name.split('(')[0][:40]:40s} {calls:5d} {tot/1e3:10.1f} us  avg {avg/1e3:10.1f} us {pct:6.2f}")
PY
find gpurun_out/prof_r2g -size +16M -delete
