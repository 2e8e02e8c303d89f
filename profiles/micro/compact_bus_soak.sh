#!/bin/bash
# ABI v7's bus formats of the node step against the oracle over other seeds (tests/test_node_step.py::test_node_step_compact_bus_parity)
mkdir -p gpurun_out
O=gpurun_out/compact_bus_soak.txt
: > $O
for s in ${SEEDS:-1000 2000 3000 4000 5000 6000 7000 8000}; do
  echo "seed offset $s: $(JG_SOAK_SEED=$s timeout 300 python -m pytest tests/test_node_step.py -m gpu -q -k compact_bus 2>&1 | tail -1)" >> $O
done
cat $O
