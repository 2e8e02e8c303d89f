#!/bin/bash
# how much of k_node_classify / k_node_route is the DISORDER of the rows: the bench's shuffle confined to windows of w partitions
export TMPDIR=/tmp
mkdir -p gpurun_out
B=$PWD/josefine_amd/host/bench_event_loop
for w in 0 1 1024 16384; do
  E=""; [ $w != 0 ] && E="JG_BENCH_SHUFFLE_WINDOW=$w"
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw_$w -o x -- $B 1000000 5 12 4 pipetasks 0 1 4 compact > gpurun_out/sw_$w.json 2>/dev/null
  echo "== window $w: $(python -c "import json;d=json.loads(open('gpurun_out/sw_$w.json').read().strip().splitlines()[-1]);print(d['ok'], '%.4g/s' % d['decisions_per_s'], d['ms_per_tick'], 'ms/tick (under the profiler)')")"
  python - gpurun_out/sw_$w/x_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("k_node_classify", "k_node_route")):
        print("  ", r["Name"].split("(")[0], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "calls", r["Calls"])
PY
  rm -rf gpurun_out/sw_$w
done
