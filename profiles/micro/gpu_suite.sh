#!/bin/bash
# the whole GPU suite, then the driver's bench command with its wall time
mkdir -p gpurun_out
timeout 2400 python -m pytest ${SUITE:-tests} -m gpu -q 2>&1 | tail -8 > gpurun_out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/gpu_suite.txt
T0=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; T1=$(date +%s); echo "bench.py default: $((T1 - T0)) s wall" >> gpurun_out/gpu_suite.txt
python - >> gpurun_out/gpu_suite.txt <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
for k,v in d['secondary'].items(): print(k, 'ERROR' if 'error' in v else {a:b for a,b in v.items() if a not in ('command','elections','round_ms_by_window')})
PY
