#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/gpu_suite.txt
python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 --repair-after 0 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/gpu_suite.txt
