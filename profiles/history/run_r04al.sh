#!/bin/bash
# late round 4: rocprofv3 showed the rows' D2H copies as __amd_rocclr_copyBuffer kernels again (27 % of GPU time) - does a kernel still pending in front of the copies (k_nop) keep them on the DMA engines?
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r04al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/$name -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none > $O/$name.json 2> $O/$name.err
  python - $name $O <<'PY'
import sqlite3, sys, json, os
name, O = sys.argv[1:3]
db = sqlite3.connect(os.path.join(O, name, "stats_results.db"))
rows = list(db.execute("select name,total_calls,total_duration,average from top_kernels"))
cp = [(n, c, t, a) for n, c, t, a in rows if "copyBuffer" in n or "fillBuffer" in n]
dec = [(c, a) for n, c, t, a in rows if "k_decode_recs" in n]
ms = json.loads(open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1])["ms_per_step"]
print(f"{name:20s} step {ms:7.3f} ms  decode_recs {dec}  blits {[(n.split('(')[0][-28:], c, round(t / 1e3, 1)) for n, c, t, a in cp]}")
PY
  rm -rf $O/$name
}
one nop AFQ_X=0
one no_nop AFQ_D2H_NOP=0
cd "$GRAFT_REPO_ROOT"
for v in 1 0 1 0; do
  AFQ_D2H_NOP=$v timeout 120 python bench.py --also none --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain run, nop=$v:', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
done
