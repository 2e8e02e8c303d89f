#!/bin/bash
# A/B: the closed loop replayed one round per graph launch (1) / several rounds per graph (default 8; 4, 16)
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'us/round %.2f' % (1e3 * d['ms_per_step']), 'value %.4g' % d['value'], 'frac %.3f' % d['roofline']['frac'])"; }
for k in 1 8 1 8 4 16 32; do
  JG_CLUSTER_ROUNDS_PER_GRAPH=$k python bench.py --cluster --steps 224 --warmup 32 --no-cpu-baseline 2>/dev/null | line closed_loop_x5_per_graph_$k
done
for k in 1 8; do
  JG_CLUSTER_ROUNDS_PER_GRAPH=$k python bench.py --cluster --replicas 3 --steps 224 --warmup 32 --no-cpu-baseline 2>/dev/null | line closed_loop_x3_per_graph_$k
  JG_CLUSTER_ROUNDS_PER_GRAPH=$k python bench.py --cluster --any-leader --replicas 3 --steps 224 --warmup 32 2>/dev/null | line any_x3_per_graph_$k
done
