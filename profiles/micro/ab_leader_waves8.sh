#!/bin/bash
# A/B: the leader half of the node tick forced to 8 waves per SIMD (-DJG_LEADER_WAVES=8: 63 VGPRs + 28 B/lane of scratch
# instead of 71 VGPRs / 7 waves): closed loop 5 x 1 M and R = 3, a = as built, b = forced
mkdir -p gpurun_out
O=gpurun_out/ab_leader_waves8.txt
: > $O
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']
print('$1', 'us/round %.2f' % (1e3 * d.get('ms_per_step_events', d['ms_per_step'])), 'frac %.3f' % r['frac'], 'leader_kernel_us %.2f' % r['leader_kernel']['avg_launch_us'])" >> $O; }
C=$PWD/josefine_amd/csrc
for v in a b a b; do
  L=$C/libjosefine_gpu.so; [ $v = b ] && L=$C/lib_w8.so.keep
  JOSEFINE_GPU_LIB=$L python bench.py --cluster --steps 224 --warmup 32 --no-cpu-baseline 2>/dev/null | line closed_loop_x5_$v
done
for v in a b; do
  L=$C/libjosefine_gpu.so; [ $v = b ] && L=$C/lib_w8.so.keep
  JOSEFINE_GPU_LIB=$L python bench.py --cluster --replicas 3 --steps 224 --warmup 32 --no-cpu-baseline 2>/dev/null | line closed_loop_x3_$v
done
cat $O
