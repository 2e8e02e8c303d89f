#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash profiles/collect.sh r01 [bench args...]
# 1) --kernel-trace --stats of the default bench command, 2) PMC passes (FETCH_SIZE,
# WRITE_SIZE in separate runs: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).
# PMC runs never combine with sys/hip/hsa tracing (pool rule).
set -u
TAG=${1:-r01}; shift || true
SUF=${JG_PROF_SUFFIX:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- \
  python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o bench -- \
    python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_pmc_$C.log 2>&1
done
find $OUT -type f | head -50
