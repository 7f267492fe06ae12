#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r02
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r02/gaptrace -o gap -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
f=$(ls $R/gpurun_out/r02/gaptrace/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $R/gpurun_out/r02/gaptrace/*/*.db | head -1)
python $R/scratch/gap_stats.py $f > $R/gpurun_out/r02/gap_stats.txt; rm -rf $R/gpurun_out/r02/gaptrace; cat $R/gpurun_out/r02/gap_stats.txt
