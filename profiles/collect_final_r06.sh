#!/bin/bash
# Round 6, final collection (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC passes.
#   bash profiles/collect_final_r06.sh [bench|stats|pmc|all]      -> gpurun_out/r06f/ (copied to profiles/r06/)
set -u
export TMPDIR=/tmp
O=gpurun_out/r06f
mkdir -p $O
part=${1:-all}
B="python bench.py"
E=josefine_amd/host/bench_event_loop
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
except Exception as e:
    print(sys.argv[1].split("/")[-1], "NO LINE", e); sys.exit(0)
r = d.get("roofline", {})
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "events", d.get("ms_per_step_events"), "frac", r.get("frac"), "launch_us", r.get("avg_launch_us"))
PY
}
if [ "$part" = all ] || [ "$part" = bench ]; then
  $B --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; line $O/bench_driver_shape.json
  $B --steps 200 --warmup 20 --no-secondary > $O/bench_1M.json 2>/dev/null; line $O/bench_1M.json
  $B --groups 4000000 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_4M.json 2>/dev/null; line $O/bench_4M.json
  $B --groups 16000000 --steps 25 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_16M.json 2>/dev/null; line $O/bench_16M.json
  $B --mode 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_1M_ragged.json 2>/dev/null; line $O/bench_1M_ragged.json
  $B --config 3 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_1250k_x3.json 2>/dev/null; line $O/bench_1250k_x3.json
  $B --cluster --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M.json 2>/dev/null; line $O/bench_cluster_1M.json
  $B --cluster --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_cluster_1M_x3.json 2>/dev/null; line $O/bench_cluster_1M_x3.json
  # configs[4] as specified: the stationary trace (vote mail / rows only), two run lengths; and rounds 2-4's trace (no repairs)
  for k in 40 200; do
    $B --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words 1 > $O/bench_routed_stationary_words_$k.json 2>/dev/null; line $O/bench_routed_stationary_words_$k.json
    $B --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words 0 > $O/bench_routed_stationary_rows_$k.json 2>/dev/null; line $O/bench_routed_stationary_rows_$k.json
  done
  $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 > $O/bench_routed_no_repairs_words.json 2>/dev/null; line $O/bench_routed_no_repairs_words.json
  $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 0 --repair-after 0 > $O/bench_routed_no_repairs_rows.json 2>/dev/null; line $O/bench_routed_no_repairs_rows.json
  $B --config 4 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_config4_share.json 2>/dev/null; line $O/bench_config4_share.json
  for L in blocked interleaved; do
    $B --cluster --any-leader --replicas 3 --leadership $L --steps 200 --warmup 20 > $O/bench_any_1M_x3_$L.json 2>/dev/null; line $O/bench_any_1M_x3_$L.json
  done
  # (round 6: the leaders of a FIVE-node cluster are elected through the transport too - 1 M elections, 40 M routed rows)
  $B --cluster --any-leader --replicas 5 --steps 200 --warmup 20 > $O/bench_any_1M_x5_blocked.json 2>/dev/null; line $O/bench_any_1M_x5_blocked.json
  $B --cluster --any-leader --replicas 5 --failures 1 --recreate --steps 60 --warmup 30 > $O/bench_any_recreate_1pct_x5_60.json 2>/dev/null; line $O/bench_any_recreate_1pct_x5_60.json
  $B --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 > $O/bench_any_failures_1pct_x3.json 2>/dev/null; line $O/bench_any_failures_1pct_x3.json
  for k in 60 240; do  # the stationary form: the groups re-created, every election won through the transport
    $B --cluster --any-leader --replicas 3 --failures 1 --recreate --steps $k --warmup 30 > $O/bench_any_recreate_1pct_x3_$k.json 2>/dev/null; line $O/bench_any_recreate_1pct_x3_$k.json
  done
  $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $O/bench_failures_1pct.json 2>/dev/null; line $O/bench_failures_1pct.json
  $B --event-loop --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_event_loop_1M.json 2> $O/bench_event_loop_1M.err; line $O/bench_event_loop_1M.json
fi
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o x -- "$@" > /dev/null 2>&1
  cp $O/stats_$name/x_kernel_stats.csv $O/kernel_stats_$name.csv 2>/dev/null
  rm -rf $O/stats_$name
}
if [ "$part" = all ] || [ "$part" = stats ]; then
  stats 1M $B --steps 100 --warmup 10 --no-cpu-baseline --no-secondary
  stats 16M $B --groups 16000000 --steps 25 --warmup 5 --no-cpu-baseline --no-secondary
  stats cluster_1M $B --cluster --steps 60 --warmup 10 --no-cpu-baseline
  stats any_1M_x3 $B --cluster --any-leader --replicas 3 --steps 100 --warmup 10
  stats routed_stationary_words $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1
  stats routed_stationary_rows $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 0
  stats routed_no_repairs_words $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0
  stats event_loop_1M_compact josefine_amd/host/bench_event_loop 1000000 5 20 5 pipetasks 0 1 4 compact
  JG_BENCH_IN_FLIGHT=2 stats event_loop_1M_compact_two_in_flight josefine_amd/host/bench_event_loop 1000000 5 20 5 pipetasks 0 1 4 compact
  stats failures_1pct $B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline
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
    pmc ${c}_16M $c $B --groups 16000000 --steps 12 --warmup 4 --no-cpu-baseline --no-secondary
    pmc ${c}_1M_ragged $c $B --mode 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
    pmc ${c}_failures $c $B --failures 1 --steps 32 --warmup 16 --no-cpu-baseline
    pmc ${c}_cluster $c $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
    pmc ${c}_any_x3 $c $B --cluster --any-leader --replicas 3 --steps 30 --warmup 10
    pmc ${c}_routed_words $c $B --cluster --failures 1 --steps 20 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0
    pmc ${c}_routed_stationary_words $c $B --cluster --failures 1 --steps 20 --warmup 10 --no-cpu-baseline --vote-words 1
  done
  for k in k_leader_tick_dense k_leader_node_tick k_follower_tick_dense k_vote_half k_votes_ k_apply_ k_route_ k_follower_slow k_cluster_claim; do KERNEL=$k python profiles/summarize_counters.py $O; done > $O/pmc_summary.txt 2>&1
  head -30 $O/pmc_summary.txt
fi
ls $O | head -80
