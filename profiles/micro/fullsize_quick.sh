#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_reference_kats.py -m gpu -q -x -k "stationary or recreate or routed_cluster_failures" 2>&1 | tail -6
