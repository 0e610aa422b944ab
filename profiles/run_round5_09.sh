#!/bin/bash
# round 5, call 9: k_pug_cell launched only when needed (first hand-back reruns its range), k_p2_search at seven workgroups per CU,
# and the whole default bench line with the new cli_sz / cli_pug legs and the compact "legs" key
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
O=$GRAFT_REPO_ROOT/gpurun_out/round5_09; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -m gpu -q -x ) > $O/tests.log 2>&1; tail -4 $O/tests.log | grep -v "^$"
( timeout 200 python tests/extended_fuzz.py 20 2000; timeout 200 python tests/extended_fuzz.py 21 1000 ) > $O/fuzz.log 2>&1; grep "extended fuzz" $O/fuzz.log; grep -c FAILED $O/fuzz.log
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; tail -c 1800 $O/bench.json; echo; tail -3 $O/bench.err
python - <<'P'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/round5_09/bench.json")))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
for k, v in d["also"].items():
    if isinstance(v, dict) and "roofline" in v: print(k, v["ms_per_step"], {a: round(b, 2) for a, b in v["roofline"]["all_kernels_ms_per_step"].items() if b > 0.3}, v.get("retries"))
    else: print(k, json.dumps(v)[:300])
P
