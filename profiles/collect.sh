#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh <tag> <suffix> [bench args...]
# 1) --kernel-trace --stats of the bench command, 2) PMC passes (FETCH_SIZE, WRITE_SIZE in
# separate runs: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2) unless JG_PROF_NO_PMC=1.
# PMC runs never combine with sys/hip/hsa tracing (pool rule).
set -u
TAG=${1:-r01}; SUF=${2:-}; shift 2 || true
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
if [ "${JG_PROF_NO_PMC:-0}" != "1" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_pmc_$C.log 2>&1
done
fi
find $OUT -type f | head -50
