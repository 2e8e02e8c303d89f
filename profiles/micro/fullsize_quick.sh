#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "any_leader" 2>&1 | tail -6
