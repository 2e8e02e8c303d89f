#!/bin/bash
# the ONE-loop modes of bench_event_loop side by side: bash profiles/micro/run_pipe_modes.sh [G] [helpers...]
mkdir -p gpurun_out
E=josefine_amd/host/bench_event_loop
G=${1:-1000000}; shift
H=${@:-4}
nproc
for m in pipe pipecolumns; do $E $G 5 16 4 $m 0 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['mode'], d['ok'], '%.3g/s' % d['decisions_per_s'], 'tick %.2f fill %.2f submit %.3f step %.2f' % (d['ms_per_tick'], d['ms_fill'], d['ms_submit'], d['ms_step_and_drain']))"; done
for h in $H; do
for m in pipetasks pipetaskscolumns; do $E $G 5 16 4 $m 0 1 $h | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['mode'], 'helpers', d['task_threads_beside_each_loop'], d['ok'], '%.3g/s' % d['decisions_per_s'], 'tick %.2f fill %.2f submit %.3f step %.2f' % (d['ms_per_tick'], d['ms_fill'], d['ms_submit'], d['ms_step_and_drain']))"; done
done
