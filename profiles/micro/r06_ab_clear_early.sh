#!/bin/bash
# Round 6 A/B: the head of a routed round - the vote mail's receiving half and the delivered rows' step - in ONE launch
# (k_round_head_multi) against one behind the other (JG_ROUTE_CLEAR_AT_HEAD=1); parity of the routed paths first
O=gpurun_out/r06_ab_clear_early
mkdir -p $O
timeout 1500 python -m pytest tests/test_dense_node.py tests/test_gpu_vote_words.py tests/test_any_leader.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3 | tee $O/parity.txt
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export JG_ROUTE_CLEAR_AT_HEAD=1; else unset JG_ROUTE_CLEAR_AT_HEAD; fi
    for k in 40 200; do
      timeout 400 python bench.py --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('clear_at_head=$v steps=$k ms/round %.4f won %s' % (d['ms_per_step'], d.get('elections_won_through_the_transport')))"
    done
    timeout 400 python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('clear_at_head=$v no repairs: ms/round %.4f' % d['ms_per_step'])"
  done
done 2>&1 | tee $O/ab.txt
unset JG_ROUTE_CLEAR_AT_HEAD
bash profiles/micro/r06_trace_round.sh > $O/trace.txt 2>&1; tail -16 $O/trace.txt
