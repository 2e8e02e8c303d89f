#!/bin/bash
# run on the GPU box: every variant library under build/ x a few grid caps
cd ${GRAFT_REPO_ROOT:-.}
for lib in build/lib_*.so; do
  for grid in ${GRIDS:-8192 2048 1024 512}; do
    JOSEFINE_GPU_LIB=$PWD/$lib JG_DENSE_GRID=$grid timeout 120 python profiles/exp_dense.py ${EXP_ARGS:-1000000 5} 2>&1 | tail -1
  done
done
