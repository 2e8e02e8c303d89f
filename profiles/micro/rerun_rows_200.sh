#!/bin/bash
mkdir -p gpurun_out/r05f
for i in 1 2; do
python bench.py --cluster --failures 1 --steps 200 --warmup 10 --no-cpu-baseline --vote-words 0 > gpurun_out/r05f/bench_routed_stationary_rows_200_run$i.json 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05f/bench_routed_stationary_rows_200_run$i.json') if l.startswith('{')][-1])
print($i, d['ms_per_step'], [round(w['ms_per_round'],4) for w in d['ms_per_round_by_leaderless_fraction']])"
done
