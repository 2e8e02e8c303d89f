#!/bin/bash
# one routed run's kernel dispatches with their timestamps (rocprofv3 --kernel-trace): what a round's launches are, in
# order, and where the stream idles between them.   bash profiles/micro/r06_trace_round.sh [extra bench flags]
mkdir -p gpurun_out/r06_trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o x -- python /root/repo/bench.py --cluster --failures 1 --steps ${STEPS:-12} --warmup 10 --no-cpu-baseline --vote-words ${WORDS:-1} "$@" > /root/repo/gpurun_out/r06_trace/line.json 2> /root/repo/gpurun_out/r06_trace/err.txt
ls -la /tmp/tr
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tr/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last ~4 rounds' worth: find the k_leader_node_tick launches and cut from the 5th last
idx = [i for i, r in enumerate(rows) if 'k_leader_node_tick' in r['Kernel_Name']]
lo = idx[-5] if len(idx) >= 5 else 0
hi = idx[-1]
out = open('/root/repo/gpurun_out/r06_trace/last_rounds.txt', 'w')
prev_end = None
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.write('%-46s dur_us %8.2f gap_us %8.2f grid %s wg %s\n' % (r['Kernel_Name'].split('(')[0][:46], (e - s) / 1e3, gap, r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?'))))
    prev_end = e
out.close()
print(open('/root/repo/gpurun_out/r06_trace/last_rounds.txt').read()[-9000:])
PY
