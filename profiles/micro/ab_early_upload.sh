#!/bin/bash
# A/B: the pipelined loop's batch uploaded when it is committed (JG_COL_UPLOAD_NOW, default) / at the head of the step
E=josefine_amd/host/bench_event_loop
G=${1:-1000000}
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['mode'], 'helpers', d['task_threads_beside_each_loop'], d['ok'], '%.3g/s' % d['decisions_per_s'], 'tick %.2f fill %.2f step %.2f' % (d['ms_per_tick'], d['ms_fill'], d['ms_step_and_drain']))"; }
for m in pipe pipetasks pipecolumns pipetaskscolumns; do
  JG_BENCH_EARLY_UPLOAD=1 $E $G 5 16 4 $m 0 1 4 | show early
  JG_BENCH_EARLY_UPLOAD=0 $E $G 5 16 4 $m 0 1 4 | show at_step
done
