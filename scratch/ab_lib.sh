#!/bin/bash
# A/B of two built libraries on one box: alternate processes.  usage: ab_lib.sh <suffix> [rounds]
cd $GRAFT_REPO_ROOT
SFX=$1; R=${2:-2}
for i in $(seq $R); do
  echo "== default"; python scratch/ab_build.py base: 2>&1 | tail -1
  echo "== $SFX"; TB_LIB_SUFFIX=$SFX python scratch/ab_build.py alt: 2>&1 | tail -1
done
