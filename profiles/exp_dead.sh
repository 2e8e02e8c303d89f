#!/bin/bash
# run on the GPU box: dense leader tick with a fraction of the groups dead (fault set), for every
# variant library under build/ — what do dead lanes in a wave cost the kernel?
#   bash profiles/exp_dead.sh            -> gpurun_out/exp_dead.log (+ PMC passes of the base library)
cd ${GRAFT_REPO_ROOT:-.}
OUT=$PWD/gpurun_out
mkdir -p $OUT
LOG=$OUT/exp_dead.log
: > $LOG
for lib in ${LIBS:-build/lib_*.so}; do
  for dead in ${DEADS:-0 0.1 0.46}; do
    JOSEFINE_GPU_LIB=$PWD/$lib DEAD_FRAC=$dead timeout 180 python profiles/exp_dense.py 1000000 5 2>&1 | tail -1 | sed "s/^/dead=$dead /" >> $LOG
  done
done
cat $LOG
if [ "${PMC:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -L > $OUT/rocprof_counters.txt 2>&1
  for dead in 0 0.46; do
    for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE" "FETCH_SIZE" \
             "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_ATOMIC_sum"; do
      tag=$(echo $C | tr ' ' '+' | cut -c1-40)
      JOSEFINE_GPU_LIB=${GRAFT_REPO_ROOT}/build/lib_BASE.so DEAD_FRAC=$dead timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv \
        -d $OUT/pmc_dead${dead}_$tag -o x -- python ${GRAFT_REPO_ROOT}/profiles/exp_dense.py 1000000 5 30 > $OUT/pmc_dead${dead}_$tag.log 2>&1
    done
  done
  python ${GRAFT_REPO_ROOT}/profiles/summarize_counters.py $OUT/pmc_dead* > $OUT/exp_dead_pmc.txt 2>&1
  cat $OUT/exp_dead_pmc.txt
fi
