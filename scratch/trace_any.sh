#!/bin/bash
# usage (on the GPU box): scratch/trace_any.sh <tag> <python script + args>  -> gpurun_out/prof_<tag>/ (rocpd db)
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -o $tag -- python $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1
ls $R/gpurun_out/prof_$tag | head
