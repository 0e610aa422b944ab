#!/bin/bash
# late round 4: why the three-step default line came out at 15.1 ms when twenty-step runs give 12.5-13: every step's own time, switch by switch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04af; mkdir -p $O
python -c "import torch" 2>/dev/null
export AFQ_BENCH_STEP_TIMES=1
run() {  # name, steps, warmup, env...
  local name=$1 st=$2 wu=$3; shift 3
  env "$@" timeout 120 python bench.py --also none --no-cpu-baseline --steps $st --warmup $wu > $O/$name.json 2> $O/$name.err
  echo "$name: $(python -c "import json,sys; print(json.loads(open('$O/$name.json').read().strip().splitlines()[-1])['ms_per_step'])") ms/step; steps: $(grep '\[bench\] step' $O/$name.err | awk '{printf "%s ", $3}')"
}
run default_3_1 3 1 AFQ_X=0
run default_8_0 8 0 AFQ_X=0
run default_8_3 8 3 AFQ_X=0
run pair_3_1 3 1 AFQ_TIMER_MODE=pair
run nochain_3_1 3 1 AFQ_CHAIN_COMPACT=0
run hosttab_3_1 3 1 AFQ_DEVICE_TABLES=0
run oldtaper_3_1 3 1 AFQ_CR_TAPER=0.28,0.56,0.78,0.92
run oldtaper_8_0 8 0 AFQ_CR_TAPER=0.28,0.56,0.78,0.92
run allold_3_1 3 1 AFQ_TIMER_MODE=pair AFQ_CHAIN_COMPACT=0 AFQ_DEVICE_TABLES=0 AFQ_CR_TAPER=0.28,0.56,0.78,0.92
run allold_8_0 8 0 AFQ_TIMER_MODE=pair AFQ_CHAIN_COMPACT=0 AFQ_DEVICE_TABLES=0 AFQ_CR_TAPER=0.28,0.56,0.78,0.92
run default_3_1b 3 1 AFQ_X=0
