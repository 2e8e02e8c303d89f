#!/bin/bash
# ABI v7's bus formats of the node step (JG_COL_PACKED_KIND, JG_NODE_COMMON_AE, JG_NODE_FSM_FUSED): parity first, then the
# event loop at 1 M x 5 with and without them, then bench.py's --event-loop line
mkdir -p gpurun_out
O=gpurun_out/compact_bus.txt
: > $O
timeout 900 python -m pytest ${TESTS:-tests/test_node_step.py tests/test_cpp_adapter.py} -m gpu -q -x 2>&1 | tail -5 >> $O
python -c "from josefine_amd.build import build_event_loop_bench as b; b()" 2>&1 | tail -1 >> $O
B=josefine_amd/host/bench_event_loop
for rep in ${REPS:-1 2}; do
  for mode in pipetasks pipetaskscolumns pipe; do
    for bus in plain compact; do
      extra=""; [ $bus = compact ] && extra="4 compact"
      timeout 300 $B 1000000 5 20 5 $mode 0 1 $extra > gpurun_out/bel_${mode}_${bus}_$rep.json 2>> $O
      python - $mode $bus $rep >> $O <<'PY'
import json, sys
m, b, r = sys.argv[1:4]
d = json.loads(open(f"gpurun_out/bel_{m}_{b}_{r}.json").read().strip().splitlines()[-1])
print(m, b, r, "ok" if d["ok"] else "FAILED", f"{d['decisions_per_s']:.4g}/s", f"{d['ms_per_tick']:.3f} ms/tick", "fill", d["ms_fill"], "submit", d["ms_submit"], "step", d["ms_step_and_drain"],
      "h2d", d["pcie_h2d_bytes_per_tick"], "d2h", d["pcie_d2h_bytes_per_tick"], "B/decision", round((d["pcie_h2d_bytes_per_tick"] + d["pcie_d2h_bytes_per_tick"]) * d["ticks"] / d["decisions"], 2))
PY
    done
  done
done
timeout 900 python bench.py --event-loop --steps 20 --warmup 5 > gpurun_out/bench_event_loop_compact.json 2> gpurun_out/bench_event_loop_compact.err
tail -c 600 gpurun_out/bench_event_loop_compact.err >> $O
python - >> $O <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/bench_event_loop_compact.json') if l.startswith('{')][-1])
ev = d["event_loop"]; tk = ev["one_loop_with_transport_and_consumer_tasks"]
print("value", d["value"], "pcie B/decision", ev["pcie_bytes_per_decision"], "tasks", tk["decisions_per_s"], tk["column_inbound_decisions_per_s"])
print("compact", {k: v for k, v in tk["compact_bus"].items() if k != "what"})
PY
cat $O
