#!/bin/bash
# Round 6 A/B: the bitmap chunk a workgroup of the vote mail's kernels takes (JG_VOTE_CHUNK words of 64 partitions: 16 / 32 = as built / 64 / 128)
O=gpurun_out/r06_ab_vote_chunk.txt
: > $O
C=$PWD/josefine_amd/csrc
for rep in 1 2; do
for c in 32 16 64 128; do
  L=$C/libjosefine_gpu.so; [ $c != 32 ] && L=$C/lib_chunk$c.so.keep
  JOSEFINE_GPU_LIB=$L timeout 400 python bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('chunk $c: ms/round %.4f won %s' % (d['ms_per_step'], d.get('elections_won_through_the_transport')))" | tee -a $O
done
done
cd /tmp && export TMPDIR=/tmp
for c in 32 64 128; do
  L=$C/libjosefine_gpu.so; [ $c != 32 ] && L=$C/lib_chunk$c.so.keep
  rm -rf /tmp/vc_$c
  JOSEFINE_GPU_LIB=$L timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vc_$c -o x -- python /root/repo/bench.py --cluster --failures 1 --steps 40 --warmup 10 --no-cpu-baseline --vote-words 1 > /dev/null 2>&1
  echo "== chunk $c" | tee -a /root/repo/$O
  python3 - /tmp/vc_$c/x_kernel_stats.csv <<'PY' | tee -a /root/repo/$O
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print('%-44s calls %5s avg_us %8.2f' % (r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
