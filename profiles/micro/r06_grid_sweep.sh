#!/bin/bash
# Round 6, review item 6: the headline kernel's grid - a persistent grid of 256 CUs x 8 waves (2048 workgroups of 4 waves:
# JG_DENSE_GRID=2048 ... 512 = 2 per CU) with its grid-stride loop against the default (one pass: ceil(G / 256) workgroups, at most 8192)
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$1', 'avg_launch_us %.2f' % r['avg_launch_us'], 'frac %.3f' % r['frac'], 'ms/step %.5f' % d['ms_per_step'], 'ceiling', r.get('stream_ceiling'))"; }
for G in 1000000 16000000; do
  K=200; [ $G = 16000000 ] && K=25
  for grid in 512 1024 2048 4096 8192 16384 65536; do
    JG_DENSE_GRID=$grid python bench.py --groups $G --steps $K --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | line "G=$G grid_cap=$grid"
  done
done
