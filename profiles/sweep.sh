#!/bin/bash
# Grid-cap / problem-size sweep of the dense kernel (run through gpurun):
#   bash profiles/sweep.sh "<grids>" "<groups>" [extra bench args]
GRIDS=${1:-"1024 2048 4096 8192 65536"}
GROUPS_=${2:-"1000000 4000000"}
shift 2 || true
for G in $GROUPS_; do for GRID in $GRIDS; do
  JG_DENSE_GRID=$GRID python bench.py --groups $G --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | \
   python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('G=$G grid=$GRID variant=${JG_DENSE_VARIANT:-0}', 'us/launch=%.2f'%r['avg_launch_us'], 'alg GB/s=%.0f'%r['achieved'], 'frac=%.3f'%r['frac'], 'dec/s=%.3e'%d['value'])"
done; done
