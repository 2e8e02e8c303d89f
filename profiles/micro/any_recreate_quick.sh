#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_any_leader.py -m gpu -x -q -k "device_parity" 2>&1 | tail -3
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'ms/round %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'decisions/round %.0f' % (d['decisions_in_timed_region']/d['steps']), 'won', d.get('elections_won_after_failures'), 'leaderless', d.get('leaderless_fraction'), 'rows/round', d.get('rows_routed_per_round'), 'appending', d.get('winners_appending_again_fraction_of_failed_groups'))"; }
for k in 60 240; do for m in 0 1; do
python bench.py --cluster --any-leader --replicas 3 --failures 1 --recreate --steps $k --warmup 30 --vote-words $m 2>gpurun_out/err_anyrec.txt | line any_x3_recreate_steps${k}_words$m
tail -2 gpurun_out/err_anyrec.txt | grep -v amdgpu.ids
done; done
