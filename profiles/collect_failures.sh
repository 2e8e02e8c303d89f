#!/bin/bash
# bash profiles/collect_failures.sh <tag>: bench line + kernel stats of `bench.py --failures 1`
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
$B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $OUT/bench_failures_1pct.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_failures -o x -- $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
cp $OUT/stats_failures/x_kernel_stats.csv $OUT/kernel_stats_failures_1pct.csv
