"""one line of josefine_amd/host/bench_event_loop's JSON, for the r06 scripts:  <binary ...> | python3 profiles/micro/el_line.py label"""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "ok", d["ok"], "in_flight", d.get("ticks_in_flight", 1), "decisions/s %.4g" % d["decisions_per_s"], "ms/tick %.3f" % d["ms_per_tick"],
      "fill %.3f submit %.3f step+outputs %.3f (in the sinks %.3f, waiting for outputs %.3f)" % (d["ms_fill"], d["ms_submit"], d["ms_step_and_drain"],
                                                                                          d.get("ms_in_the_sinks", 0), d.get("ms_waiting_for_outputs", 0)),
      "general", d["rows_general"], "B/decision %.1f" % ((d["pcie_h2d_bytes_per_tick"] + d["pcie_d2h_bytes_per_tick"]) * d["ticks"] / d["decisions"]))
