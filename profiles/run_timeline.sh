#!/bin/bash
# One-step timeline (kernels + memory copies, with their queues) of a bench command: rocprofv3 --kernel-trace --memory-copy-trace.
#   $1 = tag; $2... = bench.py flags.  Output: gpurun_out/tl_$1/ (rocpd database) + a text timeline of the last step.
TAG=${1:-tl}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/trace -o trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none "$@" > $OUT/bench.json 2> $OUT/err.txt
python $ROOT/profiles/timeline.py $OUT/trace/trace_results.db > $OUT/timeline.txt 2>&1
find $OUT -size +24M -delete
tail -5 $OUT/timeline.txt
