#!/bin/bash
# bash profiles/prof_cluster_failures.sh <tag> : bench line + kernel stats of `bench.py --cluster --failures 1`
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
$B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_cluster_failures_1pct.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cluster_failures -o x -- $B --cluster --failures 1 --steps 50 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
cp $OUT/stats_cluster_failures/x_kernel_stats.csv $OUT/kernel_stats_cluster_failures_1pct.csv
python - $OUT/kernel_stats_cluster_failures_1pct.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print(r["Name"][:84].ljust(84), r["Calls"].rjust(6), ("%.1f"%(float(r["AverageNs"])/1000)).rjust(9), r["Percentage"])
PY
