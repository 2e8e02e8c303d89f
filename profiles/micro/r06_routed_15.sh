#!/bin/bash
# Round 6: the routed round at 15 launches (the job tables' copy inside k_votes_clear, the validation on the census): parity of the
# routed paths, then configs[4] as specified both ways, the A/B with the validation pass back (JG_ROUTE_VALIDATE_PASS=1), the dispatches
O=gpurun_out/r06_routed_15
mkdir -p $O
timeout 1500 python -m pytest tests/test_dense_node.py tests/test_gpu_vote_words.py tests/test_any_leader.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4 | tee $O/parity.txt
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export JG_ROUTE_VALIDATE_PASS=1; else unset JG_ROUTE_VALIDATE_PASS; fi
    for k in 40 200; do
      timeout 400 python bench.py --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('validate_pass=$v steps=$k ms/round %.4f won %s leaderless %s' % (d['ms_per_step'], d.get('elections_won_through_the_transport'), d.get('leaderless_fraction')))"
    done
  done
done 2>&1 | tee $O/ab.txt
unset JG_ROUTE_VALIDATE_PASS
timeout 400 python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('no repairs: ms/round %.4f' % d['ms_per_step'])" | tee -a $O/ab.txt
bash profiles/micro/r06_trace_round.sh > $O/trace.txt 2>&1; tail -36 $O/trace.txt
