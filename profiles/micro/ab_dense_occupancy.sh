#!/bin/bash
# A/B: the dense halves of the closed loop at the occupancy they would have with the general state machine's registers
# the three libraries: C=josefine_amd/csrc; build as is -> cp $C/libjosefine_gpu.so $C/lib_a.so.keep; JG_LEADER_WAVES=3 JG_FOLLOWER_WAVES=4
# python -c 'from josefine_amd import build; build.build_hip(force=True)' -> lib_b.so.keep; JG_LEADER_WAVES=2 JG_FOLLOWER_WAVES=3 -> lib_c.so.keep
#   a = as built (leader 7, follower 7 waves/SIMD)   b = leader 3, follower 4   c = leader 2, follower 3
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'us/round %.2f' % (1e3 * d['ms_per_step']), 'frac %.3f' % d['roofline']['frac'])"; }
C=josefine_amd/csrc
for v in a b c a b c; do
  cp $C/lib_$v.so.keep $C/libjosefine_gpu.so; touch $C/libjosefine_gpu.so
  python bench.py --cluster --steps 224 --warmup 32 --no-cpu-baseline 2>/dev/null | line closed_loop_x5_$v
done
cp $C/lib_a.so.keep $C/libjosefine_gpu.so
