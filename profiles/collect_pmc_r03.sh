cd /tmp && export TMPDIR=/tmp
REPO=/root/repo; OUT=$REPO/gpurun_out/r03; mkdir -p $OUT
B="python $REPO/bench.py"
pmc() { local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o x -- "$@" > /dev/null 2>&1
  cp /tmp/pmc_$name/x_counter_collection.csv $OUT/pmc_$name.csv 2>/dev/null; }
pmc FETCH_SIZE_1M FETCH_SIZE $B --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_1M WRITE_SIZE $B --steps 20 --warmup 5 --no-cpu-baseline
pmc FETCH_SIZE_cluster FETCH_SIZE $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
pmc WRITE_SIZE_cluster WRITE_SIZE $B --cluster --steps 30 --warmup 10 --no-cpu-baseline
pmc FETCH_SIZE_1M_ragged FETCH_SIZE $B --mode 1 --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_1M_ragged WRITE_SIZE $B --mode 1 --steps 20 --warmup 5 --no-cpu-baseline
pmc FETCH_SIZE_4M FETCH_SIZE $B --groups 4000000 --steps 20 --warmup 5 --no-cpu-baseline
pmc WRITE_SIZE_4M WRITE_SIZE $B --groups 4000000 --steps 20 --warmup 5 --no-cpu-baseline
ls -la $OUT | grep pmc
