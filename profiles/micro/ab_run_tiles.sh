#!/bin/bash
# A/B: 256-row tiles for small delivered batches (default) vs 1024-row tiles for all (JG_ROUTE_BIG_TILES=1)
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'rows/round', d.get('rows_routed_per_round'))"; }
for i in 1 2; do
python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_x3_small_tiles
JG_ROUTE_BIG_TILES=1 python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_x3_big_tiles
done
python bench.py --cluster --any-leader --replicas 5 --failures 1 --steps 50 --warmup 10 2>/dev/null | line any_x5_small_tiles
JG_ROUTE_BIG_TILES=1 python bench.py --cluster --any-leader --replicas 5 --failures 1 --steps 50 --warmup 10 2>/dev/null | line any_x5_big_tiles
