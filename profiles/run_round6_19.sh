#!/bin/bash
# round 6, call 19: k_p2_lone on a stream of its own beside the first half of the flat build (AFQ_TEST_P2_LONE_STREAM=main: behind the search, as before)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/round6_19; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_pug.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
for V in side main side main; do
  echo "== lone stream: $V" | tee -a $O/lone.txt
  ( [ $V = main ] && export AFQ_TEST_P2_LONE_STREAM=main; python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also none --workload configs2 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms_per_step']; print(d['ms_per_step'], k.get('k_p2_lone'), k['k_p2_graph'])" | tee -a $O/lone.txt
done
