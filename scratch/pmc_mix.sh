#!/bin/bash
# usage: scratch/pmc_mix.sh <outname> <abs python script + args...>  -- instruction-mix counters (2 PMC passes)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; shift
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F16 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MUL_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$i --output-format csv -- python "$@" > $OUT.log$i 2>&1
done
