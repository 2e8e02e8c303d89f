#!/bin/bash
# Round 6 A/B on the stationary routed round (1 M x 5, vote mail): the delivered rows' step beside the receiving half on a
# side stream (JG_ROUTE_NO_SIDE_STREAM=1: one behind the other), and how many buckets a workgroup of k_route_sort_build takes.
mkdir -p gpurun_out/r06_ab
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', 'ms/round %.4f' % d['ms_per_step'], 'decisions %d' % d['decisions_in_timed_region'], 'won', d['elections_won_through_the_transport'], [round(w['ms_per_round'],4) for w in d['ms_per_round_by_leaderless_fraction']])"; }
B="python bench.py --cluster --failures 1 --steps 80 --warmup 10 --no-cpu-baseline --vote-words 1"
for rep in 1 2; do
  $B 2>/dev/null | line side_stream_sort2
  JG_ROUTE_NO_SIDE_STREAM=1 $B 2>/dev/null | line no_side_stream_sort2
  JG_ROUTE_SORT_BUCKETS=1 $B 2>/dev/null | line side_stream_sort1
  JG_ROUTE_SORT_BUCKETS=4 $B 2>/dev/null | line side_stream_sort4
done
timeout 600 python -m pytest tests/test_dense_node.py tests/test_gpu_vote_words.py -m gpu -x -q 2>&1 | tail -3
