#!/bin/bash
# A/B of the routed round (BASELINE configs[4] as specified): rows for everything (default) / the election vocabulary as
# mailbox words (JG_ROUTE_VOTE_WORDS=1, jg_votes.h).  FIRST the opt-in parity tests (the switch changes what travels, not
# what the nodes compute: every column of every node against the oracle clusters), then the lines (the decisions counted
# in the timed region must agree), then the kernels of one run of each.
#   gpurun --timeout 900 -- 'bash profiles/micro/ab_vote_words.sh > gpurun_out/ab_vote_words.txt 2>&1'
mkdir -p gpurun_out
JG_ROUTE_VOTE_WORDS=1 python -m pytest tests/test_gpu_vote_words.py -m gpu -x -q 2>&1 | tail -5
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions %d' % d['decisions_in_timed_region'], 'rows/round', d.get('rows_routed_per_round'), 'leaderless', d.get('leaderless_fraction'))"; }
for i in 1 2; do
python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/err_rows.txt | line routed_rows
JG_ROUTE_VOTE_WORDS=1 python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/err_words.txt | line routed_words
done
tail -3 gpurun_out/err_words.txt
python bench.py --cluster --failures 1 --replicas 3 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_x3_rows
JG_ROUTE_VOTE_WORDS=1 python bench.py --cluster --failures 1 --replicas 3 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_x3_words
# per-partition leadership under failures (whole groups restart, the campaigns are won through the mail)
python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('any_x3_rows', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions %d' % d['decisions_in_timed_region'], 'won', d.get('elections_won_after_failures'))"
JG_ROUTE_VOTE_WORDS=1 python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('any_x3_words', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions %d' % d['decisions_in_timed_region'], 'won', d.get('elections_won_after_failures'))"
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  JG_ROUTE_VOTE_WORDS=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vw_$m -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  echo "== kernels, JG_ROUTE_VOTE_WORDS=$m"
  python3 - /tmp/vw_$m/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print('%-60s calls %6s avg_us %9.2f total_ms %9.2f' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  cp /tmp/vw_$m/x_kernel_stats.csv /root/repo/gpurun_out/kernel_stats_vote_words_$m.csv
done
