#!/bin/bash
# usage (GPU box): scratch/refresh_profiles_r4b.sh -> gpurun_out/r06/: the default 250-step bench line and the secondary bench lines of round 6
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/default.log 2>&1; grep '^{"metric"' gpurun_out/r06/default.log | tail -1 > gpurun_out/r06/bench_default_run.json
python bench.py --precision bf16 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r06/bf16.log 2>&1; grep '^{"metric"' gpurun_out/r06/bf16.log | tail -1 > gpurun_out/r06/bench_bf16.json
TB_FORCE_DIST=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r06/dist1.log 2>&1; grep '^{"metric"' gpurun_out/r06/dist1.log | tail -1 > gpurun_out/r06/bench_force_dist_one_rank.json
python bench.py --vae --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r06/vae.log 2>&1; grep '^{"metric"' gpurun_out/r06/vae.log | tail -1 > gpurun_out/r06/bench_with_vae_encoder.json
python bench.py --batch 16 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r06/b16.log 2>&1; grep '^{"metric"' gpurun_out/r06/b16.log | tail -1 > gpurun_out/r06/bench_config5_batch16.json
python bench.py --workload sd21 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r06/sd21.log 2>&1; grep '^{"metric"' gpurun_out/r06/sd21.log | tail -1 > gpurun_out/r06/bench_config4_sd21_96.json
python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r06/fp32.log 2>&1; grep '^{"metric"' gpurun_out/r06/fp32.log | tail -1 > gpurun_out/r06/bench_fp32_mode.json
for f in gpurun_out/r06/bench_*.json; do echo $f; cut -c1-160 $f; done
