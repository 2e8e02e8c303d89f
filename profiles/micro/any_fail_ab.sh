#!/bin/bash
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'won', d.get('elections_won_after_failures'), 'rows/round', d.get('rows_routed_per_round'))"; }
for i in 1 2; do for m in 0 1; do
python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 --vote-words $m 2>/dev/null | line any_x3_failures_words$m
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/af -o x -- python /root/repo/bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 --vote-words 0 > /dev/null 2>&1
python3 - /tmp/af/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print('%-50s calls %6s avg_us %9.2f total_ms %9.2f' % (r['Name'].split('(')[0][:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
