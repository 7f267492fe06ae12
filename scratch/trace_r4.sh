#!/bin/bash
# usage (GPU box): scratch/trace_r4.sh [tag] -> gpurun_out/<tag>/: bench line, kernel-trace stats and one-replay timeline of bench.py
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/$TAG
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/trace -o $TAG -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/$TAG/trace.log 2>&1
grep '^{"metric"' $R/gpurun_out/$TAG/trace.log | tail -1 > $R/gpurun_out/$TAG/trace_bench_line.json
f=$(ls $R/gpurun_out/$TAG/trace/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $R/gpurun_out/$TAG/trace/*/*.db | head -1)
python $R/scratch/rocpd_stats.py $f 80 > $R/gpurun_out/$TAG/kernel_stats.txt
python $R/scratch/step_timeline.py $f $R/gpurun_out/$TAG/step_timeline.txt
rm -rf $R/gpurun_out/$TAG/trace
ls -la $R/gpurun_out/$TAG
