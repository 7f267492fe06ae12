#!/bin/bash
# usage (GPU box): scratch/refresh_profiles_r6.sh -> gpurun_out/r06/: kernel-trace stats of the bench command, FETCH / WRITE PMC passes, SQ counters
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r06
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/trace -o r06 -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r06/trace.log 2>&1
grep '^{"metric"' $R/gpurun_out/r06/trace.log | tail -1 > $R/gpurun_out/r06/trace_bench_line.json
f=$(ls $R/gpurun_out/r06/trace/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $R/gpurun_out/r06/trace/*/*.db | head -1)
python $R/scratch/rocpd_stats.py $f 80 > $R/gpurun_out/r06/kernel_stats.txt
python $R/scratch/step_timeline.py $f $R/gpurun_out/r06/step_timeline.txt
rm -rf $R/gpurun_out/r06/trace
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r06/pmc_traffic -o $set --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r06/pmc_$set.log 2>&1
done
python $R/scratch/pmc_traffic.py $R/gpurun_out/r06/pmc_traffic $R/gpurun_out/r06/pmc_traffic.json > $R/gpurun_out/r06/pmc_traffic_top.txt
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r06/pmc_sq -o p$i --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r06/pmc_sq$i.log 2>&1
done
python - <<PY > $R/gpurun_out/r06/pmc_sq.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/r06/pmc_sq/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:70]
        if any(x in k for x in ("gemm", "conv_halo", "attn_", "ff_fused")):
            grid = r.get("Grid_Size", "")
            if k.startswith("attn_") and "_il_" in k: k = k + " grid=" + grid   # the L0 self-attention launches have their own symbols
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc (two passes) over bench.py --no-graph: per-dispatch averages of the MFMA kernels")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(next(iter(c.values())))
    print(f"{k}  (x{n})")
    for name, v in sorted(c.items()): print(f"   {name:30s} {sum(v)/len(v):16.0f}")
PY
python $R/scratch/mfma_busy.py $R/gpurun_out/r06/pmc_sq.txt $R/gpurun_out/r06/mfma_busy.json > $R/gpurun_out/r06/mfma_busy.txt
rm -rf $R/gpurun_out/r06/pmc_sq $R/gpurun_out/r06/pmc_traffic
ls -la $R/gpurun_out/r06
