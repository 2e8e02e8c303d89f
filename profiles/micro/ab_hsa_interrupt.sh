#!/bin/bash
# A/B: completion signals by interrupt (default) / by polling (HSA_ENABLE_INTERRUPT=0) on the driver-shaped headline run
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'value %.4g' % d['value'], 'wall us/step %.3f' % (1e3*d['ms_per_step']), 'events %.3f' % (1e3*d['ms_per_step_events']), 'frac %.3f' % d['roofline']['frac'])"; }
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | line interrupt
HSA_ENABLE_INTERRUPT=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | line polling
done
python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_interrupt
HSA_ENABLE_INTERRUPT=0 python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_polling
