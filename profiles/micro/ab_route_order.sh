#!/bin/bash
# A/B of the routed round: the ordering pass launched before (default) / after (JG_ROUTE_SYNC_FIRST=1) the host has the counts
mkdir -p gpurun_out
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'rows/round', d.get('rows_routed_per_round'), 'leaderless', d.get('leaderless_fraction'))"; }
for i in 1 2; do
python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_default
JG_ROUTE_SYNC_FIRST=1 python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_sync_first
done
python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_default
JG_ROUTE_SYNC_FIRST=1 python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_sync_first
