#!/bin/bash
# Round 6: the tiled row passes - kernel stats and PMC traffic (one --pmc pass per counter) of ONE pipelined event loop at 1 M x 5, compact bus
export TMPDIR=/tmp
O=gpurun_out/r06_tiled
mkdir -p $O
B=$PWD/josefine_amd/host/bench_event_loop
timeout 600 python -m pytest tests/test_node_step.py -m gpu -x -q -k "tiled or parity or compact" 2>&1 | tail -3
for mode in tiled flat; do
  E=""; [ $mode = flat ] && E="JG_NODE_FLAT=1"
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$mode -o x -- $B 1000000 5 20 5 pipetasks 0 1 4 compact > /dev/null 2>&1
  cp $O/st_$mode/x_kernel_stats.csv $O/kernel_stats_event_loop_1M_compact_$mode.csv 2>/dev/null
  rm -rf $O/st_$mode
  for c in FETCH_SIZE WRITE_SIZE; do
    env $E rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o x -- $B 1000000 5 12 4 pipetasks 0 1 4 compact > /dev/null 2>&1
    cp $O/pmc_$c/x_counter_collection.csv $O/pmc_${c}_event_loop_1M_compact_${mode}.csv 2>/dev/null
    rm -rf $O/pmc_$c
  done
  echo "== $mode"
  python - $O $mode <<'PY'
import csv, collections, sys
O, mode = sys.argv[1], sys.argv[2]
st = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(f"{O}/kernel_stats_event_loop_1M_compact_{mode}.csv"))}
def per_kernel(path, ctr):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
f = per_kernel(f"{O}/pmc_FETCH_SIZE_event_loop_1M_compact_{mode}.csv", "FETCH_SIZE")
w = per_kernel(f"{O}/pmc_WRITE_SIZE_event_loop_1M_compact_{mode}.csv", "WRITE_SIZE")
for k in sorted(f, key=lambda k: -sum(f[k])):
    if len(f[k]) < 8 or not k.replace("void ", "").startswith("k_node"): continue
    fv, wv = f[k][len(f[k]) // 2:], w.get(k, [0])[len(w.get(k, [0])) // 2:]
    us = float(st[k]["AverageNs"]) / 1e3 if k in st else float("nan")
    # (KB per launch; the guide's gfx950 correction: FETCH_SIZE x 2)
    print(f"{k[:44]:44s} avg {us:8.1f} us  launches {len(f[k]):3d}  fetch {2 * sum(fv) / len(fv) / 1e3:9.1f} MB  write {sum(wv) / max(len(wv), 1) / 1e3:9.1f} MB")
PY
done
