#!/bin/bash
# usage (GPU box): scratch/timeline_only.sh <tag>  -> gpurun_out/<tag>_timeline.txt (one replayed step, kernel by kernel)
R=$GRAFT_REPO_ROOT; T=${1:-tl}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tl_$T -o tl -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${T}_trace.log 2>&1
f=$(ls /tmp/tl_$T/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls /tmp/tl_$T/*/*.db | head -1)
python $R/scratch/step_timeline.py $f $R/gpurun_out/${T}_timeline.txt
grep '^{"metric"' $R/gpurun_out/${T}_trace.log | tail -1 | cut -c1-200
