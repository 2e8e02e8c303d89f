#!/bin/bash
# Round 6, review item 7: one real node's dense halves by local slot numbering (profiles/micro/exp_node_layout.py), kernel stats per layout
export TMPDIR=/tmp
O=gpurun_out/r06_layout
mkdir -p $O
for layout in led_first interleaved; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$layout -o x -- python profiles/micro/exp_node_layout.py $layout 1000000 24 > $O/out_$layout.txt 2>&1
  tail -1 $O/out_$layout.txt
  cp $O/st_$layout/x_kernel_stats.csv $O/kernel_stats_node_$layout.csv 2>/dev/null
  rm -rf $O/st_$layout
  python - $O/kernel_stats_node_$layout.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if n.replace("void ", "").startswith(("k_leader_node_tick", "k_follower_tick_dense", "k_dense_slow", "k_follower_slow")):
        print("   %-40s calls %4s avg_us %8.2f" % (n[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
