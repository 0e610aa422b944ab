#!/bin/bash
# round 3: tail model with per-molecule extra refs - twin test, both tail legs
mkdir -p gpurun_out/r03s; O=gpurun_out/r03s
python -c "import torch" 2>/dev/null
timeout 200 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "tail" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for w in configs2 configs1; do
  timeout 170 python bench.py --workload $w --na-model tail --steps 2 --warmup 1 --also none --cpu-seconds 4 > $O/$w.json 2> $O/$w.err; echo "$w rc=$?"
  tail -c 1500 $O/$w.json; tail -5 $O/$w.err
done
