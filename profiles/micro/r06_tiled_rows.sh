#!/bin/bash
# Round 6: jg_step_node's row passes TILED (k_node_bin_count / _scan / _scatter + k_node_tile) against the flat ones
# (JG_NODE_FLAT=1: k_node_prefill + k_node_classify + k_node_route) - parity on the device, then ONE pipelined event loop with
# its task threads at 1 M x 5, compact bus (7 M shuffled rows per tick): decisions/s without a profiler, kernel stats with it.
export TMPDIR=/tmp
O=gpurun_out/r06_tiled
mkdir -p $O
timeout 1500 python -m pytest tests/test_node_step.py tests/test_cpp_adapter.py -m gpu -x -q 2>&1 | tail -4 | tee $O/parity.txt
B=$PWD/josefine_amd/host/bench_event_loop
for mode in tiled flat; do
  E=""; [ $mode = flat ] && E="JG_NODE_FLAT=1"
  for bus in compact plain; do
    extra="4"; [ $bus = compact ] && extra="4 compact"
    env $E $B 1000000 5 20 5 pipetasks 0 1 $extra 2>/dev/null | tail -1 > $O/line_${mode}_$bus.json
    echo "== $mode $bus: $(python -c "import json;d=json.loads(open('$O/line_${mode}_$bus.json').read().strip().splitlines()[-1]);print(d['ok'], '%.4g decisions/s' % d['decisions_per_s'], d['ms_per_tick'], 'ms/tick')")"
  done
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$mode -o x -- $B 1000000 5 20 5 pipetasks 0 1 4 compact > /dev/null 2>&1
  cp $O/st_$mode/x_kernel_stats.csv $O/kernel_stats_event_loop_1M_compact_$mode.csv 2>/dev/null
  rm -rf $O/st_$mode
  head -12 $O/kernel_stats_event_loop_1M_compact_$mode.csv | cut -d, -f1-4 | cut -c1-120
done
