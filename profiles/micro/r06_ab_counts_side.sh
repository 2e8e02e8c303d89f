#!/bin/bash
# Round 6: the routed round with the counts' copy on a side stream and the job tables' copy inside k_votes_clear (15 launches) against
# the counts inline (JG_ROUTE_COUNTS_INLINE=1); then the round's dispatches; then the event loop's first ticks with two in flight
O=gpurun_out/r06_ab_counts
mkdir -p $O
for rep in 1 2; do
  for inl in 0 1; do
    if [ $inl = 1 ]; then export JG_ROUTE_COUNTS_INLINE=1; else unset JG_ROUTE_COUNTS_INLINE; fi
    for k in 40 200; do
      timeout 400 python bench.py --cluster --failures 1 --steps $k --warmup 10 --no-cpu-baseline --vote-words 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('counts_inline=$inl steps=$k ms/round %.4f won %s leaderless %s' % (d['ms_per_step'], d.get('elections_won_through_the_transport'), d.get('leaderless_fraction')))"
    done
  done
done 2>&1 | tee $O/ab.txt
unset JG_ROUTE_COUNTS_INLINE
bash profiles/micro/r06_trace_round.sh > $O/trace.txt 2>&1; tail -45 $O/trace.txt
JG_BENCH_TICK_TRACE=1 JG_BENCH_IN_FLIGHT=2 josefine_amd/host/bench_event_loop 1000000 5 12 3 pipetasks 0 1 4 compact 2>&1 | cut -c1-220 | tee $O/ticks_w3.txt
