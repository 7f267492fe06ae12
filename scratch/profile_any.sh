#!/bin/bash
# usage (GPU box): scratch/profile_any.sh <tag> <bench args...>  -> gpurun_out/prof_<tag>/ (rocpd db) + bench JSON line
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag -- python $R/bench.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1
grep '^{"metric"' $R/gpurun_out/prof_$tag.log > $R/gpurun_out/prof_$tag.bench.json
tail -1 $R/gpurun_out/prof_$tag.bench.json | cut -c1-200
