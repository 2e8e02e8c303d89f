#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vote_words.py tests/test_dense_node.py -m gpu -x -q -k "vote_mail or stationary" 2>&1 | tail -3
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'decisions %d' % d['decisions_in_timed_region'], 'rows/round', d.get('rows_routed_per_round'))"; }
python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 2>/dev/null | line no_repairs_words
python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 2>/dev/null | line stationary_words
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mr -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1 > /dev/null 2>&1
python3 - /tmp/mr/x_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    n=r['Name'].split('(')[0]
    if 'vote' in n or 'census' in n or 'rec_multi' in n:
        print('%-44s calls %5s avg_us %8.2f' % (n[:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
