#!/bin/bash
# the stationary configs[4] trace: parity on the device, then lines both ways at two run lengths
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_node.py -m gpu -x -q -k stationary 2>&1 | tail -5
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions %d' % d['decisions_in_timed_region'], 'rows/round', d.get('rows_routed_per_round'), 'leaderless', d.get('leaderless_fraction'), [round(w['ms_per_round'], 4) for w in d['ms_per_round_by_leaderless_fraction']])"; }
for m in 0 1; do for dr in 1; do
for k in ${STEPS:-40 200}; do
timeout 300 python bench.py --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words $m --drain-applies $dr 2>gpurun_out/err_st_$m.txt | line stationary_words${m}_steps${k}_drain$dr
tail -2 gpurun_out/err_st_$m.txt
done
done
done
cd /tmp && export TMPDIR=/tmp
for m in ${PROF_MODES:-0 1}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vw_$m -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words $m > /dev/null 2>&1
  echo "== kernels, --vote-words $m"
  python3 - /tmp/vw_$m/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print('%-60s calls %6s avg_us %9.2f total_ms %9.2f' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  cp /tmp/vw_$m/x_kernel_stats.csv /root/repo/gpurun_out/kernel_stats_stationary_words_$m.csv
done
