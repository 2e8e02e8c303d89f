#!/bin/bash
# the driver-shaped default run (headline + every secondary): bash profiles/run_secondaries.sh <tag>
set -u
O=gpurun_out/r04; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape_$1.json 2> $O/bench_driver_shape_$1.err
python - <<PY
import json
d = json.load(open("$O/bench_driver_shape_$1.json"))
print("headline", d["value"], d["ms_per_step"], d["ms_per_step_events"], d["roofline"]["frac"])
s = d["secondary"]
for k, v in s.items():
    print(k, {a: b for a, b in v.items() if a not in ("command", "elections", "leaderless_fraction")})
PY
