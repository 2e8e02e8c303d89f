#!/bin/bash
# Round 4, final collection (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC passes.
#   bash profiles/collect_final_r04.sh [bench|stats|pmc|all]
set -u
export TMPDIR=/tmp
O=gpurun_out/r04f
mkdir -p $O
part=${1:-all}
B="python bench.py"
E=josefine_amd/host/bench_event_loop
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "events", d.get("ms_per_step_events"), "frac", r.get("frac"), "launch_us", r.get("avg_launch_us"))
PY
}
if [ "$part" = all ] || [ "$part" = bench ]; then
  $B --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; line $O/bench_driver_shape.json
  $B --steps 200 --warmup 20 --no-secondary > $O/bench_1M.json 2>/dev/null; line $O/bench_1M.json
  $B --groups 4000000 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_4M.json 2>/dev/null; line $O/bench_4M.json
  $B --mode 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_1M_ragged.json 2>/dev/null; line $O/bench_1M_ragged.json
  $B --config 3 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_1250k_x3.json 2>/dev/null; line $O/bench_1250k_x3.json
  $B --cluster --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M.json 2>/dev/null; line $O/bench_cluster_1M.json
  $B --cluster --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M_x3.json 2>/dev/null; line $O/bench_cluster_1M_x3.json
  $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_cluster_failures_1pct.json 2>/dev/null; line $O/bench_cluster_failures_1pct.json
  $B --config 4 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_config4_share.json 2>/dev/null; line $O/bench_config4_share.json
  for L in blocked interleaved; do
    $B --cluster --any-leader --replicas 3 --leadership $L --steps 200 --warmup 20 > $O/bench_any_1M_x3_$L.json 2>/dev/null; line $O/bench_any_1M_x3_$L.json
  done
  $B --cluster --any-leader --replicas 5 --steps 200 --warmup 20 > $O/bench_any_1M_x5_blocked.json 2>/dev/null; line $O/bench_any_1M_x5_blocked.json
  $B --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 > $O/bench_any_failures_1pct_x3.json 2>/dev/null; line $O/bench_any_failures_1pct_x3.json
  $B --cluster --any-leader --replicas 5 --failures 1 --steps 50 --warmup 10 > $O/bench_any_failures_1pct_x5.json 2>/dev/null; line $O/bench_any_failures_1pct_x5.json
  $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $O/bench_failures_1pct.json 2>/dev/null; line $O/bench_failures_1pct.json
  $B --event-loop --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_event_loop_1M.json 2> $O/bench_event_loop_1M.err; line $O/bench_event_loop_1M.json
  $B --event-loop --groups 100000 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_event_loop_100k.json 2>/dev/null; line $O/bench_event_loop_100k.json
fi
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o x -- "$@" > /dev/null 2>&1
  cp $O/stats_$name/x_kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null
  rm -rf $O/stats_$name
}
if [ "$part" = all ] || [ "$part" = stats ]; then
  stats 1M $B --steps 100 --warmup 10 --no-cpu-baseline --no-secondary
  stats cluster_1M $B --cluster --steps 60 --warmup 10 --no-cpu-baseline
  stats any_1M_x3 $B --cluster --any-leader --replicas 3 --steps 100 --warmup 10
  stats cluster_failures_1pct $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline
  stats failures_1pct $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline
  stats event_loop_100k $E 100000 5 30 5 pipe
  stats any_failures_x3 $B --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20
  head -4 $O/kernel_stats_1M.csv | cut -c1-160
fi
pmc() {  # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -o x -- "$@" > /dev/null 2>&1
  cp $O/pmc_$name/x_counter_collection.csv $O/pmc_${name}_counter_collection.csv 2>/dev/null
  rm -rf $O/pmc_$name
}
if [ "$part" = all ] || [ "$part" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc ${c}_1M $c $B --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
    pmc ${c}_4M $c $B --groups 4000000 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
    pmc ${c}_1M_ragged $c $B --mode 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
    pmc ${c}_failures $c $B --failures 1 --steps 32 --warmup 16 --no-cpu-baseline
    pmc ${c}_cluster $c $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
    pmc ${c}_any_x3 $c $B --cluster --any-leader --replicas 3 --steps 30 --warmup 10
    pmc ${c}_routed $c $B --cluster --failures 1 --steps 20 --warmup 10 --no-cpu-baseline
  done
  for k in k_leader_tick_dense k_leader_node_tick k_follower_tick_dense k_apply_vote k_route_ k_follower_slow k_cluster_claim; do KERNEL=$k python profiles/summarize_counters.py $O; done > $O/pmc_summary.txt 2>&1
  head -30 $O/pmc_summary.txt
fi
ls $O | head -80
if [ "$part" = cluster_ab ]; then
  $B --cluster --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M.json 2>/dev/null; line $O/bench_cluster_1M.json
  $B --cluster --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M_x3.json 2>/dev/null; line $O/bench_cluster_1M_x3.json
  $B --cluster --any-leader --replicas 3 --steps 200 --warmup 20 > $O/bench_any_1M_x3_blocked.json 2>/dev/null; line $O/bench_any_1M_x3_blocked.json
  for c in FETCH_SIZE WRITE_SIZE; do pmc ${c}_cluster $c $B --cluster --steps 30 --warmup 10 --no-cpu-baseline; done
  KERNEL=k_follower_tick_dense python profiles/summarize_counters.py $O | grep -A3 cluster
fi
if [ "$part" = anyfail ]; then
  stats any_failures_x3 $B --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20
  head -16 $O/kernel_stats_any_failures_x3.csv | cut -c1-150
  $B --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 > $O/bench_any_failures_1pct_x3.json 2>/dev/null; line $O/bench_any_failures_1pct_x3.json
fi
