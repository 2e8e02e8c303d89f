#!/bin/bash
# the cluster lines, short: closed loop, routed round, per-partition leadership (plain, failures)
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'events', d.get('ms_per_step_events'), 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'])"; }
python bench.py --cluster --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line closed_loop_x5
python bench.py --cluster --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line closed_loop_x3
python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_x5
python bench.py --cluster --any-leader --replicas 3 --steps 200 --warmup 20 2>/dev/null | line any_x3
python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_x3_failures
