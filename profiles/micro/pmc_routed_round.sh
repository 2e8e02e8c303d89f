cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -o x -- $B > /dev/null 2>&1
  python3 - $c <<'PY'
import csv,sys,collections
c=sys.argv[1]
v=collections.defaultdict(list)
for r in csv.DictReader(open(f'/tmp/p_{c}/x_counter_collection.csv')):
    k=r['Kernel_Name'].split('(')[0]
    if any(x in k for x in ('rec_multi','xq_multi','vote_runs','follower_slow_multi','sort_build','follower_tick_dense_multi')):
        v[k].append(float(r['Counter_Value']))
for k,a in v.items():
    t=a[len(a)//2:]
    print(c,k,'n',len(a),'mean %.1f max %.1f'%(sum(t)/len(t),max(t)))
PY
done
