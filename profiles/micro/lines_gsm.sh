#!/bin/bash
# the lines the general state machine's kernels weigh on
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'])"; }
for i in 1 2; do
  python bench.py --failures 1 --steps 160 --warmup 64 --no-cpu-baseline --no-secondary 2>/dev/null | line failures_tick
  python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_x5
  python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_x3_failures
done
