#!/bin/bash
# bash profiles/pmc_node.sh <tag>: SQ counters of k_leader_node_tick (exp_node.py) next to the ack-only dense tick
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc_node_$1 -o x -- python $REPO/profiles/exp_node.py 1000000 5 30 > /dev/null 2>&1
  cp $OUT/pmc_node_$1/x_counter_collection.csv $OUT/pmc_node_$1.csv 2>/dev/null
}
pass SQ1 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM"
pass SQ2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
# (a FETCH_SIZE / WRITE_SIZE pass of this script did not finish in 10 minutes under counter collection: not collected)
python $REPO/profiles/summarize_counters.py $OUT/pmc_node_SQ1.csv $OUT/pmc_node_SQ2.csv $OUT/pmc_node_MEM.csv 2>/dev/null | grep -i "node_tick\|==" 
