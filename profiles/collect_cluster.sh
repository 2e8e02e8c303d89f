#!/bin/bash
# bash profiles/collect_cluster.sh <tag>: the closed-loop lines and kernel stats (with and without failures)
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
$B --cluster --steps 100 --warmup 20 > $OUT/bench_cluster_1M.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cluster_1M -o x -- $B --cluster --steps 100 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
cp $OUT/stats_cluster_1M/x_kernel_stats.csv $OUT/kernel_stats_cluster_1M.csv
bash $REPO/profiles/prof_cluster_failures.sh $TAG > /dev/null
