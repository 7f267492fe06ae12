#!/bin/bash
# usage: scratch/pmc_lds.sh <outname> <python script + args>  -- LDS / wait counters
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$1; shift
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT -o p$i --output-format csv -- python "$@" > $OUT.log$i 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(x in r["Kernel_Name"] for x in ("gemm", "conv", "attn")):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()): print(f"   {n:28s} {sum(v)/len(v):14.0f}  (x{len(v)})")
PY
