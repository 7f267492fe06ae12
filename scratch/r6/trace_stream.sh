#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6; cd /tmp; export TMPDIR=/tmp
for bits in 39 1048615; do
  rm -rf /tmp/ts_$bits
  rocprofv3 --kernel-trace --stats -d /tmp/ts_$bits -o t -- python $R/scratch/r6/trace_stream.py $bits > /tmp/ts_$bits.log 2>&1
  f=$(ls /tmp/ts_$bits/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls /tmp/ts_$bits/*/*.db | head -1)
  echo "== bits $bits"; python $R/scratch/rocpd_stats.py $f 12 | cut -c1-130
done
