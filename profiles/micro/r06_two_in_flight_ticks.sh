#!/bin/bash
# per-tick host trace of ONE loop with two ticks in flight (JG_BENCH_TICK_TRACE), then the device timeline of the same
B=$PWD/josefine_amd/host/bench_event_loop
O=$PWD/gpurun_out/r06_two_in_flight
mkdir -p $O
nproc; lscpu | grep -i "model name\|numa node(s)" | head -3
for rep in 1 2; do
JG_BENCH_TICK_TRACE=1 JG_BENCH_IN_FLIGHT=2 $B 1000000 5 14 6 pipetasks 0 1 4 compact 2>&1 | cut -c1-200 | tail -9
done
bash profiles/micro/r06_two_in_flight_trace.sh > $O/trace_run.txt 2>&1
grep "no profiler" $O/trace_run.txt
sed -n 1,75p $O/timeline_inflight2.txt
