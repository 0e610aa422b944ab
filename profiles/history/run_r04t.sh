#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -c "import torch" 2>/dev/null
AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_timing.so timeout 300 python bench.py --workload configs2 --na-model tail --steps 1 --warmup 0 --also none --no-cpu-baseline > gpurun_out/r04t_clocks.txt 2> gpurun_out/r04t_clocks.err
grep "em2 hybrid" gpurun_out/r04t_clocks.txt | sort -t= -k2 -n -r | awk 'NR%2==1' | head -14
python3 /dev/stdin gpurun_out/r04t_clocks.txt 9 <<'PY'
import re,collections,sys
launches=[[]]
for l in open(sys.argv[1]):
    if l.startswith('em2 blk end'): launches.append([]); continue
    m=re.match(r"em2 blk (\w+) (\d+) (\d+) (\d+)",l)
    if m:
        k=int(m[1],16); launches[-1].append((k>>56, int(m[2]),int(m[3]),int(m[4])))
for L in launches:
    for tier in range(6):
        B=[r for r in L if r[0]==tier]
        if not B: continue
        body=[(r[3]-r[1])/100 for r in B]
        print(f"tier {tier}: {len(B)} blocks, mean {sum(body)/len(body):.0f} us, max {max(body):.0f}, sum {sum(body)/1e3:.0f} ms")
    print('--')
PY
