#!/bin/bash
# Round 6, review item 4: two ticks in flight in BatchedEventLoop (JG_NODE_KEEP, ABI v9).  Parity first (the node step's two
# in flight against the oracle's synchronous steps; the cluster of loops against the oracle-backed loops), then ONE loop with
# its transport / consumer tasks at 1 M x 5, compact bus and plain, one and two ticks in flight, interrupt and polled waits.
#   bash profiles/micro/r06_two_in_flight.sh   -> gpurun_out/r06_two_in_flight/
O=gpurun_out/r06_two_in_flight
mkdir -p $O
timeout 1500 python -m pytest tests/test_node_step.py tests/test_cpp_adapter.py -m gpu -x -q -k "two_in_flight or pipelined or compact_bus_event or async_parity" 2>&1 | tail -8 | tee $O/parity.txt
python -c "
from josefine_amd.build import build_event_loop_bench
print(build_event_loop_bench())"
B=josefine_amd/host/bench_event_loop
line() { python3 profiles/micro/el_line.py "$1"; }
for rep in 1 2; do
for f in 1 2; do
  for mode in pipetasks pipetaskscolumns; do
    JG_BENCH_IN_FLIGHT=$f timeout 300 $B 1000000 5 40 10 $mode 0 1 4 compact 2> $O/err.txt | tee $O/${mode}_compact_inflight$f.json | line ${mode}_compact_inflight$f
    tail -2 $O/err.txt
  done
  JG_BENCH_IN_FLIGHT=$f timeout 300 $B 1000000 5 40 10 pipetasks 0 1 4 2> $O/err.txt | tee $O/pipetasks_plain_inflight$f.json | line pipetasks_plain_inflight$f
  JG_BENCH_IN_FLIGHT=$f HSA_ENABLE_INTERRUPT=0 timeout 300 $B 1000000 5 40 10 pipetasks 0 1 4 compact 2> $O/err.txt | tee $O/pipetasks_compact_polled_inflight$f.json | line pipetasks_compact_polled_inflight$f
  JG_BENCH_IN_FLIGHT=$f timeout 300 $B 1000000 5 40 10 pipe 0 1 4 compact 2> $O/err.txt | tee $O/pipe_compact_inflight$f.json | line pipe_one_thread_compact_inflight$f
done
done
JG_BENCH_IN_FLIGHT=2 JG_TRACE_NODE=1 timeout 300 $B 1000000 5 12 4 pipetasks 0 1 4 compact 2>&1 | tail -14 > $O/trace_node.txt
