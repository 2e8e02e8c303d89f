#!/bin/bash
mkdir -p gpurun_out
JG_BENCH_TRACE_ROUNDS=1 python bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 2> gpurun_out/trace_norepair.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], [round(w['ms_per_round'],3) for w in d['ms_per_round_by_leaderless_fraction']])"
grep '\[bench\] round' gpurun_out/trace_norepair.txt | tail -18
