#!/bin/bash
# A/B: the general state machine's kernels as the compiler sizes them (a) / held to 128 VGPRs = 4 waves per SIMD (b)
# the two libraries: C=josefine_amd/csrc; build as is -> cp $C/libjosefine_gpu.so $C/lib_a.so.keep; JG_GSM_WAVES=4
# python -c 'from josefine_amd import build; build.build_hip(force=True)' -> lib_b.so.keep (josefine_amd/build.py passes the macro on)
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'])"; }
C=josefine_amd/csrc
for v in a b a b; do
  cp $C/lib_$v.so.keep $C/libjosefine_gpu.so; touch $C/libjosefine_gpu.so
  python bench.py --failures 1 --steps 160 --warmup 64 --no-cpu-baseline --no-secondary 2>/dev/null | line failures_tick_$v
  python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_x5_$v
  python bench.py --cluster --any-leader --replicas 3 --failures 1 --steps 100 --warmup 20 2>/dev/null | line any_x3_failures_$v
done
cp $C/lib_a.so.keep $C/libjosefine_gpu.so
