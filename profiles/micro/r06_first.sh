#!/bin/bash
# Round 6, first light of the transport's new order (phase, emission index, sender): the routed paths' parity tests on the
# device, then configs[4] as specified (the stationary trace WITHOUT synthetic votes) both ways, per-partition leadership at
# R = 5 with its leaders ELECTED through the transport, and the kernels of one routed run.
#   bash profiles/micro/r06_first.sh        -> gpurun_out/r06_first/
mkdir -p gpurun_out/r06_first
O=gpurun_out/r06_first
timeout 1500 python -m pytest tests/test_dense_node.py tests/test_gpu_vote_words.py tests/test_any_leader.py -m gpu -x -q 2>&1 | tail -6 > $O/parity.txt
cat $O/parity.txt
line() { python -c "
import json,sys
try:
    d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
except Exception as e:
    print('$1', 'NO LINE', e); sys.exit(0)
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions', d.get('decisions_in_timed_region'), 'rows/round', d.get('rows_routed_per_round'), 'leaderless', d.get('leaderless_fraction'), 'won', d.get('elections_won_through_the_transport', d.get('elections_won_after_failures')), 'elections', d.get('elections'))"; }
for m in 1 0; do
  for k in 40 200; do
    timeout 400 python bench.py --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words $m 2> $O/err_st_$m.txt | tee $O/bench_routed_stationary_words${m}_$k.json | line stationary_words${m}_steps${k}
    tail -2 $O/err_st_$m.txt
  done
done
timeout 400 python bench.py --cluster --any-leader --replicas 5 --steps 200 --warmup 20 2> $O/err_any5.txt | tee $O/bench_any_1M_x5_elected.json | line any_x5_elected
tail -2 $O/err_any5.txt
timeout 400 python bench.py --cluster --any-leader --replicas 5 --failures 1 --recreate --steps 60 --warmup 30 2> $O/err_any5f.txt | tee $O/bench_any_recreate_1pct_x5.json | line any_x5_recreate
tail -2 $O/err_any5f.txt
timeout 400 python bench.py --cluster --any-leader --replicas 3 --failures 1 --recreate --steps 60 --warmup 30 2> $O/err_any3f.txt | tee $O/bench_any_recreate_1pct_x3.json | line any_x3_recreate
tail -2 $O/err_any3f.txt
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vw_$m -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words $m > /dev/null 2>&1
  echo "== kernels, --vote-words $m"
  python3 - /tmp/vw_$m/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print('%-60s calls %6s avg_us %9.2f total_ms %9.2f' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  cp /tmp/vw_$m/x_kernel_stats.csv /root/repo/$O/kernel_stats_routed_stationary_words_$m.csv
done
