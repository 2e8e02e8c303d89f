#!/bin/bash
# rocprofv3 kernel stats of ONE pipelined event loop with its task threads at 1 M x 5, plain and compact bus
export TMPDIR=/tmp
mkdir -p gpurun_out
B=$PWD/josefine_amd/host/bench_event_loop
for bus in plain compact; do
  extra="4"; [ $bus = compact ] && extra="4 compact"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$bus -o x -- $B 1000000 5 20 5 pipetasks 0 1 $extra > gpurun_out/st_$bus.json 2> gpurun_out/st_$bus.err
  cp gpurun_out/st_$bus/x_kernel_stats.csv gpurun_out/kernel_stats_event_loop_1M_$bus.csv 2>/dev/null
  rm -rf gpurun_out/st_$bus
  echo "== $bus"; head -14 gpurun_out/kernel_stats_event_loop_1M_$bus.csv | cut -c1-150
done
