#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_dense_node.py tests/test_any_leader.py tests/test_gpu_vote_words.py tests/test_cpp_adapter.py -m gpu -q -x 2>&1 | tail -4
line() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d['roofline']; lk=r.get('leader_kernel') or {}
print('$1', 'us/round %.2f' % (d.get('ms_per_step_events', d['ms_per_step'])*1e3), 'frac %.3f' % r['frac'], 'leader_us', lk.get('avg_launch_us'), 'leader_frac', lk.get('frac'))"; }
python bench.py --cluster --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line closed_loop_x5
python bench.py --cluster --replicas 3 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line closed_loop_x3
python bench.py --cluster --any-leader --replicas 3 --steps 200 --warmup 20 2>/dev/null | line any_x3_blocked
python bench.py --cluster --any-leader --replicas 3 --leadership interleaved --steps 200 --warmup 20 2>/dev/null | line any_x3_interleaved
python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | line routed_words
