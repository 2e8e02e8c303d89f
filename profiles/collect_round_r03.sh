#!/bin/bash
# Collect round 3's evidence on the GPU box (run through gpurun):  bash profiles/collect_round_r03.sh
# Bench lines (JSON) of every configuration, rocprofv3 --kernel-trace --stats of the modes that changed this
# round, per-launch percentiles of the failure tick, and the PMC passes (each --pmc run on its own, kernel
# trace only: pool rule).  Outputs: gpurun_out/r03/ (copy what is to be judged into profiles/r03/).
TAG=r03
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
E=$REPO/josefine_amd/host/bench_event_loop
# ---- bench lines
$B --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2>/dev/null
$B --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_gpus2_self_launched_aliased.json 2>/dev/null
$B --groups 4000000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_4M.json 2>/dev/null
$B --groups 1250000 --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1250k_x3.json 2>/dev/null
$B --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1M_x5_ragged.json 2>/dev/null
$B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $OUT/bench_failures_1pct.json 2>/dev/null
$B --cluster --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cluster_1M.json 2>/dev/null
JG_CLUSTER_SEPARATE_HALVES=1 $B --cluster --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_cluster_1M_separate_halves.json 2>/dev/null
$B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cluster_failures_1pct.json 2>/dev/null
JG_ROUTE_SEPARATE_LAUNCHES=1 JG_ROUTE_LIBRARY_SORT=1 JG_CLUSTER_SEPARATE_HALVES=1 JG_ROUTE_NO_VOTES_KERNEL=1 JG_ROUTE_ROW_PER_LANE=1 $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cluster_failures_1pct_round2_config.json 2>/dev/null
$B --event-loop --groups 100000 --loops 2,4 > $OUT/bench_event_loop_100k.json 2>/dev/null
$B --event-loop > $OUT/bench_event_loop_1M.json 2>/dev/null   # 1 M x 5, one loop and the best of 4 / 8 loops
$B --event-loop --groups 10000 --replicas 3 --loops 1 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_event_loop_10k_x3.json 2>/dev/null
$REPO/profiles/micro/exp_pcie > $OUT/exp_pcie.log 2>&1
# ---- kernel stats
prof() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o x -- "$@" > /dev/null 2>&1
  cp $OUT/stats_$name/x_kernel_stats.csv $OUT/kernel_stats_$name.csv 2>/dev/null
}
prof 1M $B --steps 100 --warmup 10 --no-cpu-baseline
prof cluster_1M $B --cluster --steps 100 --warmup 20 --no-cpu-baseline
prof cluster_failures_1pct $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline
prof event_loop_100k $E 100000 5 50 10 inplace
rocprofv3 --kernel-trace --output-format csv -d $OUT/stats_failures -o x -- $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
python - $OUT/stats_failures/x_kernel_trace.csv > $OUT/failure_tick_kernel_percentiles.txt <<'PY'
# per-launch durations of the failure tick's kernels: median / p95 / min / max over the TIMED launches (the last 160 of each)
import csv, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel, launches, median_us, p95_us, min_us, max_us  (bench.py --failures 1 --steps 160 --warmup 64; the last 160 launches of each kernel)")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v[-160:])
    if len(v) < 20: continue
    print(f"{k}, {len(v)}, {v[len(v)//2]:.2f}, {v[int(len(v)*0.95)]:.2f}, {v[0]:.2f}, {v[-1]:.2f}")
PY
# ---- PMC
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$name -o x -- "$@" > /dev/null 2>&1
  cp $OUT/pmc_$name/x_counter_collection.csv $OUT/pmc_$name.csv 2>/dev/null
}
pmc FETCH_SIZE_1M FETCH_SIZE $B --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_1M WRITE_SIZE $B --steps 20 --warmup 5 --no-cpu-baseline
pmc FETCH_SIZE_cluster FETCH_SIZE $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
pmc WRITE_SIZE_cluster WRITE_SIZE $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
pmc FETCH_SIZE_event_loop FETCH_SIZE $E 100000 5 20 5 inplace
pmc WRITE_SIZE_event_loop WRITE_SIZE $E 100000 5 20 5 inplace
cd $REPO
for k in k_leader_tick_dense k_leader_node_tick k_follower_tick_dense k_node_; do KERNEL=$k python profiles/summarize_counters.py $OUT/pmc_*/; done > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/stats_* $OUT/pmc_*/
ls $OUT
