#!/bin/bash
# usage (on the GPU box): scratch/profile_bench.sh <tag>  -> gpurun_out/prof_<tag>/ (rocpd db) + bench JSON line
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$1 -o $1 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_$1.log 2>&1
grep '^{"metric"' $R/gpurun_out/prof_$1.log > $R/gpurun_out/prof_$1.bench.json
tail -1 $R/gpurun_out/prof_$1.bench.json | cut -c1-160
