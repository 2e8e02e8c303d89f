#!/bin/bash
# Round 6: JG_NODE_KEEP's parity test over other seeds on the device (4 cases per seed), and the cluster of loops at depth 2 at other sizes
O=gpurun_out/r06_two_in_flight
mkdir -p $O
for s in 1 2 3 5 8 13 21 34 55 89; do
  echo "seed $s: $(JG_SOAK_SEED=$s timeout 600 python -m pytest tests/test_node_step.py -m gpu -q -k two_in_flight 2>&1 | tail -1)"
done 2>&1 | tee $O/soak_two_in_flight_seeds.txt
python - <<'PY' 2>&1 | tee -a $O/soak_two_in_flight_seeds.txt
import sys; sys.path.insert(0, 'tests')
import test_cpp_adapter as t
dev, ora = t.build_cluster_test(oracle=False), t.build_cluster_test(oracle=True)
for args in ((100_000, 5, 50, "scripted"), (20_000, 3, 60, "scripted"), (5000, 3, 150, "elect"), (2000, 5, 200, "elect"), (1000, 3, 150, "failover")):
    for env in ({"JG_CLUSTER_PIPELINED": "1", "JG_CLUSTER_IN_FLIGHT": "2"}, {"JG_CLUSTER_PIPELINED": "1", "JG_CLUSTER_IN_FLIGHT": "2", "JG_CLUSTER_COMPACT": "1"}):
        try:
            a, b = t.run_cluster(dev, *args, env=env), t.run_cluster(ora, *args, env=env)
            print(args, sorted(env), "EQUAL" if a == b else "DIFFER", a[:150])
        except AssertionError as e:
            print(args, sorted(env), "FAILED", str(e)[:300])
PY
