cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none"
one() { $B "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_ms_per_step']; print(d['ms_per_step'], d['value'], {x:round(k[x],2) for x in k if k[x]>0.3})"; }
for i in 1 2; do
echo base; AFQ_LIB_PATH=$GRAFT_REPO_ROOT/alevin-fry_amd/csrc/libafquant_base.so one --workload configs2
echo new; one --workload configs2
done
timeout 600 python -m pytest tests/test_gpu_pug.py -x -q 2>&1 | tail -2
