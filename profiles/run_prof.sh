#!/bin/bash
# Collect the rocprofv3 evidence for one bench.py command (run on the GPU box through gpurun).
#   $1 = tag (e.g. r2a); $2... = extra bench.py flags (e.g. --workload configs2).  Outputs land under
#   gpurun_out/prof_$1/ and are summarised into profiles/ by summarize.py / traffic.py.
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
# PASSES (env) picks the passes: stats sq sq2 fetch write tcc (default: all).
set -u
TAG=${1:-r2}
shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
PASSES=${PASSES:-"stats sq sq2 fetch write tcc"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also none $*"
BENCH1="python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --also none $*"
echo "$BENCH" > $OUT/command.txt
for P in $PASSES; do
  case $P in
    stats) timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.err ;;
    sq)    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_sq -o pmc -- $BENCH1 > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err ;;
    sq2)   timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $BENCH1 > /dev/null 2> $OUT/pmc_sq2.err ;;
    fetch) timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err ;;
    write) timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_write.err ;;
    tcc)   timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o pmc -- $BENCH > /dev/null 2> $OUT/pmc_tcc.err ;;
  esac
done
# keep the merged payload small: drop per-dispatch traces bigger than 16 MB
find $OUT -size +16M -delete
ls -la $OUT
