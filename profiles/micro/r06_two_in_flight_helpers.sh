#!/bin/bash
# Round 6: two ticks in flight - how many transport / consumer task threads beside the loop's (the box has 16 CPUs)
#   bash profiles/micro/r06_two_in_flight_helpers.sh  -> gpurun_out/r06_two_in_flight/helpers.txt
O=gpurun_out/r06_two_in_flight
mkdir -p $O
B=josefine_amd/host/bench_event_loop
nproc
for h in 4 6 8 11; do
  for f in 1 2; do
    JG_BENCH_IN_FLIGHT=$f timeout 300 $B 1000000 5 40 10 pipetasks 0 1 $h compact 2>/dev/null | python3 profiles/micro/el_line.py helpers_${h}_interrupt
    JG_BENCH_IN_FLIGHT=$f HSA_ENABLE_INTERRUPT=0 timeout 300 $B 1000000 5 40 10 pipetasks 0 1 $h compact 2>/dev/null | python3 profiles/micro/el_line.py helpers_${h}_polled
  done
done 2>&1 | tee $O/helpers.txt
