#!/bin/bash
# Collect one round's evidence on the GPU box (run through gpurun):  bash profiles/collect_round.sh r02
# Bench lines (JSON) of every configuration, rocprofv3 --kernel-trace --stats of the three modes, and the
# PMC passes (each --pmc run on its own, kernel trace only: pool rule).  Outputs: gpurun_out/<tag>/.
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
# ---- bench lines
$B --steps 200 --warmup 20 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2>/dev/null
$B --groups 4000000 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_4M.json 2>/dev/null
$B --groups 1250000 --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1250k_x3.json 2>/dev/null
$B --groups 10000 --replicas 3 --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_10k_x3_ragged.json 2>/dev/null
$B --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1M_x5_ragged.json 2>/dev/null
$B --groups 1250000 --replicas 3 --mode 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_1250k_x3_ragged.json 2>/dev/null
$B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $OUT/bench_failures_1pct.json 2>/dev/null
$B --cluster --steps 100 --warmup 20 > $OUT/bench_cluster_1M.json 2>/dev/null
$B --single-process --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_single_process_1.json 2>/dev/null
$B --single-process --alias-devices --gpus 2 --groups 500000 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_single_process_2x500k_aliased.json 2>/dev/null
# ---- kernel stats
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o x -- $B "$@" > /dev/null 2>&1
  cp $OUT/stats_$name/x_kernel_stats.csv $OUT/kernel_stats_$name.csv 2>/dev/null
}
prof 1M --steps 100 --warmup 10 --no-cpu-baseline
prof 4M --groups 4000000 --steps 100 --warmup 10 --no-cpu-baseline
prof 1M_ragged --mode 1 --steps 100 --warmup 10 --no-cpu-baseline
prof failures_1pct --failures 1 --steps 160 --warmup 64 --no-cpu-baseline
prof cluster_1M --cluster --steps 100 --warmup 20 --no-cpu-baseline
# ---- PMC
pmc() {  # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$name -o x -- $B "$@" > /dev/null 2>&1
  cp $OUT/pmc_$name/x_counter_collection.csv $OUT/pmc_$name.csv 2>/dev/null
}
pmc FETCH_SIZE_1M FETCH_SIZE --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_1M WRITE_SIZE --steps 20 --warmup 5 --no-cpu-baseline
pmc FETCH_SIZE_1M_ragged FETCH_SIZE --mode 1 --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_1M_ragged WRITE_SIZE --mode 1 --steps 20 --warmup 5 --no-cpu-baseline
pmc SQ_1M "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" --steps 20 --warmup 5 --no-cpu-baseline
pmc SQ_failures "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" --failures 1 --steps 100 --warmup 10 --no-cpu-baseline
pmc LDS_failures "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_FLAT" --failures 1 --steps 100 --warmup 10 --no-cpu-baseline
pmc FETCH_SIZE_failures FETCH_SIZE --failures 1 --steps 100 --warmup 10 --no-cpu-baseline
pmc WRITE_SIZE_failures WRITE_SIZE --failures 1 --steps 100 --warmup 10 --no-cpu-baseline
pmc FETCH_SIZE_cluster FETCH_SIZE --cluster --steps 30 --warmup 10 --no-cpu-baseline
pmc WRITE_SIZE_cluster WRITE_SIZE --cluster --steps 30 --warmup 10 --no-cpu-baseline
cd $REPO
for k in k_leader_tick_dense k_leader_node_tick k_follower_tick_dense; do KERNEL=$k python profiles/summarize_counters.py $OUT/pmc_*; done > $OUT/pmc_summary.txt 2>&1
python profiles/exp_compact.py > $OUT/exp_compact.log 2>&1
python profiles/exp_node.py > $OUT/exp_node.log 2>&1
python profiles/exp_fail_census.py > $OUT/exp_fail_census.log 2>&1
rm -rf $OUT/stats_* $OUT/pmc_*/  # keep the summaries (the raw trees are large)
ls $OUT
