#!/bin/bash
# Round 6: bench.py --event-loop as profiles/r06/bench_event_loop_1M.json is made (1 M x 5, loops 4,8), with the two-ticks-in-flight figures
O=gpurun_out/r06_two_in_flight
mkdir -p $O
time (timeout 1500 python bench.py --event-loop --steps 20 --warmup 5 --loops 4,8 > $O/bench_event_loop_1M.json 2> $O/bench_event_loop_1M.err)
tail -3 $O/bench_event_loop_1M.err
python3 - $O/bench_event_loop_1M.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
ev = d['event_loop']; tk = ev['one_loop_with_transport_and_consumer_tasks']; cb = tk['compact_bus']; two = cb['two_ticks_in_flight']
print('value %.4g (loops %d)' % (d['value'], ev['loops']))
print('one loop + tasks: plain %.4g  compact %.4g (%.3f ms/tick)  compact columns %.4g' % (tk['decisions_per_s'], cb['decisions_per_s'], cb['ms_per_tick'], cb['column_inbound_decisions_per_s']))
print('two ticks in flight: compact %.4g (%.3f ms/tick, %s)  polled %.4g  columns %.4g  plain %.4g  PCIe %.1f GB/s both ways' % (two['decisions_per_s'], two['ms_per_tick'], two['ms_per_tick_parts'], two['polled_decisions_per_s'], two['column_inbound_decisions_per_s'], two['plain_bus_decisions_per_s'], two['pcie_GB_per_s_both_ways']))
PY
time (timeout 600 python bench.py --event-loop --steps 12 --warmup 3 --loops 4 > $O/bench_event_loop_driver_shape.json 2>/dev/null); tail -c 600 $O/bench_event_loop_driver_shape.json | head -c 300
