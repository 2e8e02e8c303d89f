#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -x -k "event_loop" 2>&1 | tail -5
timeout 900 python bench.py --event-loop --steps 12 --warmup 3 --loops 4 --no-cpu-baseline > gpurun_out/bench_event_loop_1M.json 2>gpurun_out/el.err; tail -3 gpurun_out/el.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_event_loop_1M.json') if l.startswith('{')][-1])
tk=d['event_loop']['one_loop_with_transport_and_consumer_tasks']
print('value', d['value'], 'tasks', tk['decisions_per_s'], tk['column_inbound_decisions_per_s'])
print('host_wait', tk['host_wait']['interrupt_decisions_per_s'], tk['host_wait']['polled_decisions_per_s'])
print('wire', {k:v for k,v in tk['wire_decode'].items() if k!='what'})
"
