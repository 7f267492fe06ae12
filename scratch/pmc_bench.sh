#!/bin/bash
# HBM traffic counters for the kernels of bench.py: separate --pmc passes, kernel-trace only (no other trace domains)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_bench
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o $set --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $OUT.$set.log 2>&1
done
ls $OUT | head
