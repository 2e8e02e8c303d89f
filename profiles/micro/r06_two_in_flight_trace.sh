#!/bin/bash
# Round 6: where a tick of the event loop goes on the DEVICE with one and two ticks in flight - rocprofv3 kernel + memory-copy
# trace of ONE loop with tasks, compact bus, 1 M x 5; a merged timeline of a few steady-state ticks (copies with their sizes,
# the kernels of a tick as one span).
#   bash profiles/micro/r06_two_in_flight_trace.sh   -> gpurun_out/r06_two_in_flight/timeline_inflight{1,2}.txt
O=$PWD/gpurun_out/r06_two_in_flight
mkdir -p $O
B=$PWD/josefine_amd/host/bench_event_loop
cd /tmp && export TMPDIR=/tmp
for f in 1 2; do
  JG_BENCH_IN_FLIGHT=$f timeout 300 $B 1000000 5 30 8 pipetasks 0 1 4 compact 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('in_flight $f (no profiler): decisions/s %.4g ms/tick %.3f fill %.3f step+outputs %.3f of which in the sinks %.3f, waiting for outputs %.3f' % (d['decisions_per_s'], d['ms_per_tick'], d['ms_fill'], d['ms_step_and_drain'], d['ms_in_the_sinks'], d['ms_waiting_for_outputs']))"
  rm -rf /tmp/tl_$f
  JG_BENCH_IN_FLIGHT=$f timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl_$f -o x -- $B 1000000 5 16 6 pipetasks 0 1 4 compact > /tmp/tl_$f.out 2>&1
  tail -1 /tmp/tl_$f.out | cut -c1-400
  python3 - /tmp/tl_$f <<'PY' > $O/timeline_inflight$f.txt
import csv, sys, glob, os
d = sys.argv[1]
kf = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
mf = glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True)
ev = []
for r in csv.DictReader(open(kf[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'].split('(')[0][:48], 0))
rows = list(csv.DictReader(open(mf[0]))) if mf else []
if rows: print('# memory copy columns:', list(rows[0].keys()))
for r in rows:
    b = next((int(r[k]) for k in ('Bytes', 'Size', 'bytes') if k in r and r[k]), 1 << 20)
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', r.get('Name', r.get('Operation', '?'))), b))
ev.sort()
# the last ~5 ticks: from the 5th last k_leader_node_tick on
lead = [i for i, e in enumerate(ev) if e[2] == 'K' and 'k_leader_node_tick' in e[3]]
i0 = lead[-6] if len(lead) >= 6 else 0
t0 = ev[i0][0]
for s, e, k, name, b in ev[i0:]:
    if k == 'C' and b < 4096: continue
    print('%9.1f us  +%8.1f us  %s %-50s %s' % ((s - t0) / 1e3, (e - s) / 1e3, k, name, ('%.1f MB' % (b / 1e6)) if k == 'C' else ''))
PY
  head -70 $O/timeline_inflight$f.txt
done
