#!/bin/bash
# Round 4 collection, part 1: GPU suite, per-partition-leadership cluster (kernel stats + bench lines).
# Run on the GPU box from the repo root: bash profiles/collect_round_r04.sh [part]
set -u
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
part=${1:-all}
if [ "$part" = all ] || [ "$part" = tests ]; then
  (timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > $O/gputests.log
  tail -5 $O/gputests.log
fi
if [ "$part" = fullsize ]; then
  timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "any_leader" 2>&1 | tail -15
fi
if [ "$part" = all ] || [ "$part" = any ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_any -o any -- python bench.py --cluster --any-leader --replicas 3 --steps 100 --warmup 10 > $O/prof_any.json 2>/dev/null
  cp $(find $O/prof_any -name "*kernel_stats.csv" | head -1) $O/kernel_stats_any_1M_x3.csv
  head -12 $O/kernel_stats_any_1M_x3.csv
  python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 > $O/bench_any_failures_1pct_x3.json 2> $O/bench_any_fail.err
  tail -c 800 $O/bench_any_fail.err
  python - <<PY
import json
d = json.load(open("$O/bench_any_failures_1pct_x3.json"))
print(d["ms_per_step"], d["roofline"]["frac"], d.get("leaderless_fraction"), d.get("failed_fraction"), d.get("elections_won_after_failures"), d.get("rows_routed_per_round"), d.get("rows_left_for_the_host"))
PY
fi
if [ "$part" = routed ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_routed -o x -- python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_cluster_failures_1pct.json 2>/dev/null
  cp $(find $O/prof_routed -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cluster_failures_1pct.csv
  head -22 $O/kernel_stats_cluster_failures_1pct.csv | cut -c1-150
  python -c "
import json; d=json.load(open('$O/bench_cluster_failures_1pct.json')); print(d['ms_per_step'], d['roofline']['frac'], d.get('ms_per_round_by_leaderless_fraction'))"
fi
