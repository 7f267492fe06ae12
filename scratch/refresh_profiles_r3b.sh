#!/bin/bash
# usage (GPU box): scratch/refresh_profiles_r3b.sh -> gpurun_out/r03/: the secondary bench lines of round 3 and the configs[4] (B=16 + fp8 attention) capture:
# rocprofv3 kernel stats of that run and the FETCH / WRITE PMC passes of its attention kernels
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r03
python bench.py --batch 16 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r03/b16.log 2>&1; grep '^{"metric"' gpurun_out/r03/b16.log | tail -1 > gpurun_out/r03/bench_config5_batch16.json
python bench.py --batch 16 --fp8-attn --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r03/b16f8.log 2>&1; grep '^{"metric"' gpurun_out/r03/b16f8.log | tail -1 > gpurun_out/r03/bench_config5_batch16_fp8_attention.json
python bench.py --workload sd21 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r03/sd21.log 2>&1; grep '^{"metric"' gpurun_out/r03/sd21.log | tail -1 > gpurun_out/r03/bench_config4_sd21_96.json
python bench.py --vae --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r03/vae.log 2>&1; grep '^{"metric"' gpurun_out/r03/vae.log | tail -1 > gpurun_out/r03/bench_with_vae_encoder.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/c5trace -o c5 -- python $R/bench.py --batch 16 --fp8-attn --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/r03/c5trace.log 2>&1
f=$(ls $R/gpurun_out/r03/c5trace/*.db 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $R/gpurun_out/r03/c5trace/*/*.db | head -1)
python $R/scratch/rocpd_stats.py $f 40 > $R/gpurun_out/r03/config5_fp8_kernel_stats.txt
rm -rf $R/gpurun_out/r03/c5trace
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r03/c5pmc -o $set --output-format csv -- python $R/bench.py --batch 16 --fp8-attn --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r03/c5pmc_$set.log 2>&1
done
python $R/scratch/pmc_traffic.py $R/gpurun_out/r03/c5pmc $R/gpurun_out/r03/config5_fp8_pmc_traffic.json > $R/gpurun_out/r03/config5_fp8_pmc_traffic_top.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/r03/c5sq -o sq --output-format csv -- python $R/bench.py --batch 16 --fp8-attn --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/r03/c5sq.log 2>&1
python - <<PY > $R/gpurun_out/r03/config5_fp8_pmc_sq.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/r03/c5sq/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:70]
        if "attn" in k or "fp8" in k:
            agg[k + " grid=" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc over bench.py --batch 16 --fp8-attn --no-graph: per-dispatch averages of the attention kernels (incl. the e4m3 P.V forward and its V-image pre-pass)")
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(next(iter(c.values())))
    print(f"{k}  (x{n})")
    for name, v in sorted(c.items()): print(f"   {name:30s} {sum(v)/len(v):16.0f}")
    b, m = c.get("SQ_BUSY_CYCLES"), c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if b and m: print(f"   MFMA-busy {100 * (sum(m)/len(m)) / (1024 * (sum(b)/len(b)) / 32):.1f} %")
PY
rm -rf $R/gpurun_out/r03/c5pmc $R/gpurun_out/r03/c5sq
ls -la $R/gpurun_out/r03
