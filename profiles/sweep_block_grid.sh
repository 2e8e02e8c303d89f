for B in 64 128 256; do
  JG_BLOCK=$B python -c "from josefine_amd.build import build_hip; build_hip(force=True)" > /dev/null 2>&1
  for GRID in 4096 8192 16384 65536; do
    for G in 1000000; do
      JG_DENSE_GRID=$GRID python bench.py --groups $G --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('block=$B grid=$GRID G=$G', 'us/launch=%.2f'%r['avg_launch_us'], 'dec/s=%.3e'%d['value'], 'T16 us=%.2f'%(d['batched_ticks']['ms_per_step']*1e3))"
    done
  done
done
python -c "from josefine_amd.build import build_hip; build_hip(force=True)" > /dev/null 2>&1
