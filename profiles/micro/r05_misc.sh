#!/bin/bash
# round 5: the 8-rank rehearsal, the failure tick standalone x3 against the default line's secondary, the 16 M x 5 headline
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bench_contract.py -m gpu -q -k "rehearsal" 2>&1 | tail -5
for i in 1 2 3; do
python bench.py --failures 1 --steps 96 --warmup 32 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('failures_tick standalone', 'tick_ms %.4f' % d['ms_per_step'], 'dense_kernel_us %.2f' % d['roofline']['avg_launch_us'], d['roofline'].get('launches_timed'))"
done
python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['secondary'].items(): print(k, {a:b for a,b in v.items() if a not in ('command','elections')})
"
python bench.py --groups 16000000 --steps 25 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_16M.json 2>gpurun_out/bench_16M.err; tail -2 gpurun_out/bench_16M.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_16M.json') if l.startswith('{')][-1])
print('16M', d['ms_per_step'], d['roofline'])
"
