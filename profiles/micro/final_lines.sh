#!/bin/bash
# the bench lines that carry roofline.traffic, after traffic.json was written from the PMC passes of the same source
O=gpurun_out/r04f; mkdir -p $O
B="python bench.py"
$B --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
$B --steps 200 --warmup 20 --no-secondary > $O/bench_1M.json 2>/dev/null
$B --groups 4000000 --steps 100 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_4M.json 2>/dev/null
$B --mode 1 --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench_1M_ragged.json 2>/dev/null
$B --failures 1 --steps 160 --warmup 64 --no-cpu-baseline > $O/bench_failures_1pct.json 2>/dev/null
for f in driver_shape 1M 4M 1M_ragged failures_1pct; do python - $O/bench_$f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.4g" % d["value"], "ms/step %.5f" % d["ms_per_step"], "frac %.3f" % r["frac"], "traffic", (r.get("traffic") or {}).get("bytes"))
PY
done
