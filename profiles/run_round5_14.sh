#!/bin/bash
# round 5, call 14: the EM legs' envelope gate and the pool re-grow change, before the evidence run
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_14; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_pug.py -m gpu -q -x -k "pool or outgrows or collision or hand or skew or narrow" ) > $O/tests.log 2>&1; tail -3 $O/tests.log | grep -v "^$"
timeout 300 python bench.py --workload configs2 --steps 2 --warmup 1 --cpu-seconds 6 --also none > $O/c2.json 2> $O/c2.err; tail -2 $O/c2.err
timeout 300 python bench.py --workload configs2 --na-model tail --steps 2 --warmup 1 --cpu-seconds 4 --also none > $O/c2t.json 2> $O/c2t.err; tail -2 $O/c2t.err
python - <<'P'
import json, os
for n in ("c2", "c2t"):
    try:
        d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], f"gpurun_out/round5_14/{n}.json")))
        c = d["cpu_baseline"]
        print(n, d["ms_per_step"], c["value"], c["sample"][:60])
        print("  em_arithmetic", {k: v for k, v in c["em_arithmetic"].items() if k not in ("what", "gate")})
        print("  em_order_sensitivity", c.get("em_order_sensitivity"))
    except Exception as e:
        print(n, "FAILED", type(e).__name__, e)
P
