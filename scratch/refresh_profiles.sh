#!/bin/bash
# usage (GPU box): scratch/refresh_profiles.sh  -> default bench run, rocprofv3 kernel stats of the same command, FETCH/WRITE PMC passes
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | tail -1 > gpurun_out/bench_default.json
cp gpurun_out/bench_kernel_table.json gpurun_out/bench_default_kernel_table.json
scratch/profile_bench.sh r01final
scratch/pmc_bench.sh > /dev/null 2>&1
tail -1 gpurun_out/bench_default.json | cut -c1-400
