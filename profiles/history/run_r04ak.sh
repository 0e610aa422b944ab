#!/bin/bash
# late round 4: rocprofv3 showed the rows' D2H copies as __amd_rocclr_copyBuffer kernels again (27 % of GPU time) - which switch brings them back?
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/r04ak; mkdir -p $O
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
one default AFQ_X=0
one shell_env GPU_FORCE_BLIT_COPY_SIZE=0
one no_chain AFQ_CHAIN_COMPACT=0
one old_taper AFQ_CR_TAPER=0.28,0.56,0.78,0.92
one old_all AFQ_CR_TAPER=0.28,0.56,0.78,0.92 AFQ_CHAIN_COMPACT=0 AFQ_TIMER_MODE=pair AFQ_DEVICE_TABLES=0
