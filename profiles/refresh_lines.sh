#!/bin/bash
# bash profiles/refresh_lines.sh <tag>: every bench line of profiles/<round>/ again (no profiler)
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
$B --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2>/dev/null
$B --groups 4000000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_4M.json 2>/dev/null
$B --groups 1250000 --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1250k_x3.json 2>/dev/null
$B --groups 10000 --replicas 3 --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_10k_x3_ragged.json 2>/dev/null
$B --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1M_x5_ragged.json 2>/dev/null
$B --groups 1250000 --replicas 3 --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1250k_x3_ragged.json 2>/dev/null
$B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $OUT/bench_failures_1pct.json 2>/dev/null
$B --cluster --steps 100 --warmup 20 > $OUT/bench_cluster_1M.json 2>/dev/null
$B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cluster_failures_1pct.json 2>/dev/null
$B --single-process --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_single_process_1.json 2>/dev/null
$B --single-process --alias-devices --gpus 2 --groups 500000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_single_process_2x500k_aliased.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_failures -o x -- $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
cp $OUT/stats_failures/x_kernel_stats.csv $OUT/kernel_stats_failures_1pct.csv
ls $OUT/*.json | wc -l
