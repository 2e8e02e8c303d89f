#!/bin/bash
# the process on SEVERAL pipelined loops with their task threads, plain / compact bus (1 M x 5 over L loops)
mkdir -p gpurun_out
O=gpurun_out/compact_loops.txt
: > $O
B=josefine_amd/host/bench_event_loop
for L in 1 2 4; do
  for bus in plain compact; do
    for mode in pipetasks pipetaskscolumns; do
      extra="4"; [ $bus = compact ] && extra="4 compact"
      Q=$((2 * L)); [ $Q -lt 4 ] && Q=4
      GPU_MAX_HW_QUEUES=$Q timeout 300 $B 1000000 5 20 5 $mode 0 $L $extra > gpurun_out/loops_${mode}_${bus}_$L.json 2>> $O
      python - $mode $bus $L >> $O <<'PY'
import json, sys
m, b, L = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/loops_{m}_{b}_{L}.json").read().strip().splitlines()[-1])
print(m, b, "loops", L, "ok" if d["ok"] else "FAILED", f"{d['decisions_per_s']:.4g}/s", f"{d['ms_per_tick']:.3f} ms/tick", "B/decision", round((d["pcie_h2d_bytes_per_tick"] + d["pcie_d2h_bytes_per_tick"]) * d["ticks"] / d["decisions"], 2))
PY
    done
  done
done
cat $O
