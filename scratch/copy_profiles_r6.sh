#!/bin/bash
# gpurun_out/r06/ (scratch, merged back from the GPU box) -> profiles/r06_* (tracked)
cd "$(dirname "$0")/.."
S=gpurun_out/r06; D=profiles
cp $S/bench_default_run.json $D/r06_bench_default_run.json
cp $S/trace_bench_line.json $D/r06_bench_trace_run.json
cp $S/kernel_stats.txt $D/r06_bench_kernel_stats.txt
cp $S/step_timeline.txt $D/r06_step_timeline.txt
cp $S/pmc_traffic.json $D/r06_pmc_traffic.json
cp $S/pmc_sq.txt $D/r06_pmc_sq.txt
cp $S/mfma_busy.txt $D/r06_mfma_busy.txt
cp $S/mfma_busy.json $D/r06_mfma_busy.json
for n in bf16 force_dist_one_rank with_vae_encoder config5_batch16 config4_sd21_96 fp32_mode; do cp $S/bench_$n.json $D/r06_bench_$n.json; done
ls -la $D/r06_*
