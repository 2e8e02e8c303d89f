#!/bin/bash
# quick loop for the vote mail: parity tests, one line each way, the kernel table of the words run
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vote_words.py -m gpu -x -q 2>&1 | tail -5
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions %d' % d['decisions_in_timed_region'], 'rows/round', d.get('rows_routed_per_round'), 'leaderless', d.get('leaderless_fraction'), 'won', d.get('elections_won_after_failures'))"; }
for m in ${MODES:-0 1}; do
timeout 300 python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words $m 2>gpurun_out/err_words_$m.txt | line routed_x5_words$m
done
for m in ${MODES:-0 1}; do
timeout 300 python bench.py --cluster --failures 1 --replicas 3 --steps 50 --warmup 10 --no-cpu-baseline --vote-words $m 2>gpurun_out/err_x3_$m.txt | line routed_x3_words$m
tail -5 gpurun_out/err_x3_$m.txt
done
cd /tmp && export TMPDIR=/tmp
for m in ${PROF_MODES:-1}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vw_$m -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words $m > /dev/null 2>&1
  echo "== kernels, --vote-words $m"
  python3 - /tmp/vw_$m/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print('%-60s calls %6s avg_us %9.2f total_ms %9.2f' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  cp /tmp/vw_$m/x_kernel_stats.csv /root/repo/gpurun_out/kernel_stats_vote_words_$m.csv
done
