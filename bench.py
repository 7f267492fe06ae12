#!/usr/bin/env python
"""bench.py -- TextBoost train steps/sec on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]      N > 1: this process launches N ranks itself (torch.distributed.run on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
                                                          (an existing launch: ranks from RANK / LOCAL_RANK / WORLD_SIZE; N must match)

A "step" is one full optimizer step of train_textboost.py:1040-1149 on synthetic 4x64x64 latents: text-encoder
LoRA fwd, frozen SD1.5 UNet fwd + dgrad bwd, MSE + KPL (teacher fwd + student fwd), text-encoder bwd, all-reduce of
the trainable gradients (N>1), GradScaler/clip/AdamW/renorm -- captured in a HIP graph, inputs resident in HBM.
Rank 0 prints ONE JSON line (fields per the driver contract, plus `roofline` and `cpu_baseline`)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 1.76e12  # UNet fwd 803.3 + dgrad bwd ~888 + CLIP 66.7 GFLOP
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, MI355X_MICROARCH.md


def roofline_leg(step):
    """One extra EAGER step with a HIP-event pair around every kernel-family launch (same stream), after the timed region.  `roofline` is
    about the single kernel symbol with the largest summed duration (what rocprofv3's per-kernel statistics can be held against);
    `roofline.classes` splits the same step into conv / linear / attention / norm with each class's distance from its roof, and
    `roofline.dominant_family` / `worst_big_family` let every family compete, the multi-kernel entry points included."""
    from textboost_amd import ops
    torch.cuda.synchronize()
    world, force = step.world, step.force_dist
    step.world, step.force_dist = 1, False  # rank 0 runs this leg alone: no collective may be issued here
    # The eager launches come out of Python ~20 us apart, shorter than most kernels' gaps: with an idle GPU an event pair would also time the
    # host gap between `record` and the launch (a 17 us GEMM read 33 us).  A spin kernel queued first keeps the stream busy while the whole
    # step is enqueued behind it, so every event pair brackets its kernel back to back on the device, as in the graph replay.
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    torch.cuda._sleep(2_000_000)
    t1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 2_000_000 / max(t0.elapsed_time(t1), 1e-3)
    ops.start_recording()
    try:
        torch.cuda._sleep(int(cycles_per_ms * 600.0))  # ~0.6 s: an eager step with events is enqueued in ~0.2 s of host time
        # what an event pair reads with NOTHING between its two records, under the same saturated-queue conditions (the command processor's
        # cost of the second timestamp): subtracted from every launch below, so that the averages agree with rocprofv3's kernel durations
        empties = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for a_, b_ in empties:
            a_.record()
            b_.record()
        step.step_eager()
        torch.cuda.synchronize()
    finally:
        rec = ops.stop_recording()
        step.world, step.force_dist = world, force
    pair_overhead_ms = sorted(a_.elapsed_time(b_) for a_, b_ in empties)[len(empties) // 2]
    agg = {}
    for name, flops, byts, e0, e1 in rec:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += max(e0.elapsed_time(e1) - pair_overhead_ms, 1e-4) * 1e-3
        a[2] += flops
        a[3] += byts
    # the roofline object is about ONE kernel symbol: candidates are the single-kernel records (the attention-backward and
    # GroupNorm entry points launch 2-3 kernels per call and are listed in the table only)
    single = {k: v for k, v in agg.items() if k.startswith(("gemm_kernel<", "conv_halo_kernel<", "gemm8_kernel<", "gemm_f32_kernel")) or k == "attn_fwd_kernel"}
    dom = max(single.items(), key=lambda kv: kv[1][1])
    name, (n, t, fl, _) = dom
    table = {k: {"launches": v[0], "total_ms": round(v[1] * 1e3, 3), "avg_us": round(v[1] / v[0] * 1e6, 2),
                 "tflops": round(v[2] / v[1] / 1e12, 1) if v[2] else None,
                 "gbps": round(v[3] / v[1] / 1e9, 1) if v[3] else None} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    achieved = fl / t / 1e12
    total_flop = sum(v[2] for v in agg.values())
    by = agg[name][3]  # algorithmic bytes of the same launches (operands read once, output written once)
    HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
    hbm_bound = by > 0 and (by / (HBM_PEAK_GBS * 1e9)) > (fl / (MFMA_PEAK_TFLOPS * 1e12))  # which roof takes longer for this work
    roof = {"bound": "mfma", "kernel": name, "launches_per_step": n, "avg_launch_us": round(t / n * 1e6, 2),
            "alg_flop_per_launch": fl / n, "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "recorded_matmul_tflop_per_step": round(total_flop / 1e12, 3), "event_pair_overhead_us": round(pair_overhead_ms * 1e3, 2)}  # sum of 2MNK / attention FLOP over the step's launches
    if by > 0:
        roof["alg_bytes_per_launch"] = round(by / n)
        roof["mfma_frac"] = roof["frac"]
        roof["hbm_frac"] = round(by / t / 1e9 / HBM_PEAK_GBS, 4)
    if hbm_bound:  # the launches' arithmetic intensity is below the machine balance (312 FLOP/B): quote the memory roof
        roof.update({"bound": "hbm", "achieved": round(by / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof["hbm_frac"]})
    # HBM traffic per launch of that kernel: PMC passes cannot run inside this process (separate rocprofv3 --pmc runs, FETCH_SIZE and
    # WRITE_SIZE, FETCH doubled per MI355X_MICROARCH.md); the committed summary of those passes (scratch/pmc_bench.sh ->
    # scratch/pmc_traffic.py) is quoted when it has the same kernel symbol
    for pf in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pf)))
            cands = [name] + ([name[:-1] + s_ for s_ in (", 0>", ", 0, 1>", ", 0, 1, false>", ", 0, 1, true>")] if name.endswith(">") else [])   # (the trace's symbol carries the
            key = next((c for c in cands if c in pmc), None)                                                   # defaulted SUB / KH / ST template arguments)
            if key:
                roof["traffic"] = round(pmc[key]["hbm_bytes_per_launch"])
                roof["traffic_source"] = f"profiles/{pf} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per launch)"
                break
        except (OSError, ValueError, KeyError):
            pass
    roof["table"] = table  # per kernel family of the same eager step: launches, total ms, avg us, TFLOP/s or GB/s (algorithmic)
    # ---- the step by CLASS (same eager leg): where the time is and how far each class sits from the roof that bounds it
    classes = {}
    for k, v in agg.items():
        c = kernel_class(k)
        a = classes.setdefault(c, [0, 0.0, 0.0, 0.0])
        for i in range(4):
            a[i] += v[i]
    roof["classes"] = {}
    for c, (n_, t_, fl_, by_) in sorted(classes.items(), key=lambda kv: -kv[1][1]):
        e = {"launches": n_, "ms": round(t_ * 1e3, 3)}
        if c in ("conv", "linear", "attention"):
            e.update({"bound": "mfma", "alg_tflop": round(fl_ / 1e12, 3), "achieved": round(fl_ / t_ / 1e12, 1), "unit": "TFLOP/s",
                      "frac": round(fl_ / t_ / 1e12 / MFMA_PEAK_TFLOPS, 4)})
        else:
            e.update({"bound": "hbm", "alg_gb": round(by_ / 1e9, 3), "achieved": round(by_ / t_ / 1e9, 1), "unit": "GB/s",
                      "frac": round(by_ / t_ / 1e9 / HBM_PEAK_GBS, 4)})
        roof["classes"][c] = e
    # every family competes here, also the entry points that launch 2-3 kernels per call (the attention backward, GroupNorm): the family with the
    # largest summed duration, and the WORST one among those that take >= 1 ms of the step -- `kernel` above stays the top single rocprof symbol
    def fam(k, v):
        mf = kernel_class(k) in ("conv", "linear", "attention")
        ach = (v[2] / v[1] / 1e12) if mf else (v[3] / v[1] / 1e9)
        return {"family": k, "launches_per_step": v[0], "total_ms": round(v[1] * 1e3, 3), "bound": "mfma" if mf else "hbm",
                "achieved": round(ach, 1), "unit": "TFLOP/s" if mf else "GB/s", "peak": MFMA_PEAK_TFLOPS if mf else HBM_PEAK_GBS,
                "frac": round(ach / (MFMA_PEAK_TFLOPS if mf else HBM_PEAK_GBS), 4)}
    fams = [fam(k, v) for k, v in agg.items() if v[1] > 0 and (v[2] > 0 or v[3] > 0)]
    roof["dominant_family"] = max(fams, key=lambda f: f["total_ms"])
    big = [f for f in fams if f["total_ms"] >= 1.0]
    roof["worst_big_family"] = min(big, key=lambda f: f["frac"]) if big else None
    return roof, table


def kernel_class(name):
    """conv | linear | attention | norm for a launch record of the eager leg (ops._rec names = the rocprofv3 kernel symbols / entry points)"""
    if name.startswith("gemm8_kernel<"):
        return "conv" if name.split(",")[4].strip() == "true" else "linear"
    if name.startswith("conv_halo_kernel<"):
        return "conv"
    if name.startswith("gemm_kernel<"):
        return "conv" if name.split(",")[2].strip() == "1" else "linear"
    if name.startswith(("lin320_kernel", "ff_fused_kernel", "gemm_f32_kernel", "gemm")):
        return "linear"
    if name.startswith("attn_"):
        return "attention"
    if name.startswith(("groupnorm_", "layernorm_")):
        return "norm"
    return "other"


def sustained_mfma_peak(ms_target=15.0, reps=3):
    """The dense-fp16 matrix rate this chip SUSTAINS (tb_mfma_peak_probe: every SIMD issuing independent v_mfma_f32_32x32x16_f16 back to
    back, two waves per SIMD, random operands), median of `reps` launches of ~`ms_target` ms: under nothing but matrix work the part holds
    ~1.5-1.75 GHz, not its 2.4 GHz peak clock, so no kernel can reach the 2.5 PFLOP/s of MI355X_MICROARCH.md for longer than microseconds."""
    from textboost_amd import _lib as L
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    blocks = 2 * cus
    sink = torch.empty(blocks * 256, device="cuda")
    flop = lambda it: blocks * 4 * it * 16 * 32768.0  # noqa: E731
    iters = 2000
    rates = []
    for i in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        L.check(L.lib().tb_mfma_peak_probe(L.ptr(sink), blocks, iters, L.stream()), "tb_mfma_peak_probe")
        b.record()
        torch.cuda.synchronize()
        ms = max(a.elapsed_time(b), 1e-3)
        if i == 0:
            iters = max(200, int(iters * ms_target / ms))  # first launch sizes the rest
        else:
            rates.append(flop(iters) / ms / 1e9)
    rates.sort()
    return round(rates[len(rates) // 2], 1)


def dominant_tile_clock(batch=8, reps=20):
    """The dominant convolution tile against its own cycle counter: one ResnetBlock2D convolution of the metric's first UNet level (3x3, 320 -> 320,
    64 x 64 maps: 256 tiles of gemm8_kernel<4, 2, 4, 5, true, 3>) on N(0,1) operands, HIP-event time per launch beside the s_memtime stamps of
    workgroup 0 (tb_gemm8_debug: kernel start [0] .. last epilogue pass [3]; shader cycles).  cycles / time = the clock the chip holds INSIDE the
    launch; 16.3 cycles x 1.25 K MFMAs per SIMD / cycles = the matrix pipe's duty in cycles.  On real operands the launch runs at ~1.6-1.75 GHz, not
    at the 2.4 GHz the 2.5 PFLOP/s peak is quoted at: the tile is power-limited (profiles/r06_ldsmfma_probe.txt)."""
    from textboost_amd import ops, _lib as L
    H, C = 64, 320
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(batch * H * H, C, device="cuda", generator=g).to(L.half_dtype())
    w = (torch.randn(C, 9 * C, device="cuda", generator=g) / (9 * C) ** 0.5).to(L.half_dtype())
    out = torch.empty(batch * H * H, C, device="cuda", dtype=L.half_dtype())
    geo = dict(B=batch, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    for _ in range(3):
        ops.gemm(x, w, out, conv=geo)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.gemm(x, w, out, conv=geo)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / reps * 1e3
    dbg = torch.zeros(32, dtype=torch.int64, device="cuda")
    L.lib().tb_gemm8_debug(L.ptr(dbg))
    try:
        ops.gemm(x, w, out, conv=geo)
        torch.cuda.synchronize()
    finally:
        L.lib().tb_gemm8_debug(None)
    d = dbg.tolist()
    cycles, loop = d[3] - d[0], d[2] - d[1]
    if cycles <= 0:
        return None
    mfma = 16.3 * 1.25 * 9 * C   # per SIMD: 256 x 160 x K / (16 x 16 x 32) / 4 MFMAs of 16.3 cycles
    return {"launch": f"conv3x3 {C}->{C} @ {H}x{H}, B={batch}", "us": round(us, 2), "cycles_workgroup0": cycles, "main_loop_cycles": loop,
            "clock_ghz_inside_launch": round(cycles / (us * 1e3), 3), "mfma_cycles": round(mfma), "mfma_duty_main_loop": round(mfma / max(loop, 1), 3),
            "tflops": round(2.0 * batch * H * H * C * 9 * C / us / 1e6, 1)}


def mfma_busy_table():
    """Per kernel family MFMA-pipe busy fraction from the committed SQ counter pass of the same bench command (profiles/rNN_mfma_busy.json:
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); PMC passes cannot run inside this process).  None when no file matches."""
    for pf in ("r06_mfma_busy.json", "r05_mfma_busy.json", "r04_mfma_busy.json", "r03_mfma_busy.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", pf)))
            return {"source": f"profiles/{pf}", "families": d}
        except (OSError, ValueError):
            pass
    return None


def cpu_baseline_leg(max_seconds=260.0, n_steps=10, b1_seconds=60.0):
    """oracle/ (the eager-PyTorch fp32 restatement of the reference step) timed on this box's host cores:
    (1) 7 to `n_steps` optimizer steps at B=1 of the same SD1.5 shapes (config 1's batch), the first two discarded as warm-up, MEDIAN of
    the rest (at least five), the steps beyond the seventh bounded by `b1_seconds`; (2) when the budget allows, ONE real step at the metric's batch 8 -- the like-for-like number.
    `value` is the real B=8 step when it ran, else the B=1 median scaled by 8 (`scaled` says which)."""
    from oracle import train_step as ts
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd.workload import synthetic_ids
    torch.manual_seed(0)

    def fast_init(m):
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() >= 2:
                    fan = p[0].numel()
                    p.uniform_(-fan ** -0.5, fan ** -0.5)
                elif "norm" in n or "ln" in n:
                    p.fill_(1.0 if n.endswith("weight") else 0.0)
                else:
                    p.zero_()
        return m
    with torch.device("meta"):
        unet = UNet2DCondition(UNetConfig.sd15())
    unet = fast_init(unet.to_empty(device="cpu"))
    base = TextBoostEncoder(CLIPTextCfg.sd15(), r=0)
    with torch.no_grad():
        null = base.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    base.set_null_embedding(null)
    teacher = ts.make_teacher(base)
    te = TextBoostEncoder(CLIPTextCfg.sd15(), r=4)
    te.load_state_dict(base.state_dict(), strict=False)
    te.set_null_embedding(null)
    added = add_tokens(te, list(range(1000, 1018)))
    st = ts.TrainState(te, teacher, unet, added, ts.StepConfig())
    g = torch.Generator().manual_seed(1)

    def one(B):
        ids = synthetic_ids(B, added, g)
        pids = synthetic_ids(B, added, g, prior=True)
        x0, noise = torch.randn(B, 4, 64, 64, generator=g), torch.randn(B, 4, 64, 64, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        t0 = time.perf_counter()
        st.step(x0, noise, t, ids, pids)
        return time.perf_counter() - t0

    times = []
    t_begin = time.perf_counter()
    for i in range(n_steps):
        times.append(one(1))
        if time.perf_counter() - t_begin > b1_seconds and len(times) >= 7:   # at least 5 timed steps behind the two warm-up ones
            break
    skip = 2 if len(times) >= 4 else 1
    timed = sorted(times[skip:])
    med = timed[len(timed) // 2]  # median of the steps after the warm-up ones
    b8 = None
    if (time.perf_counter() - t_begin) + 7.0 * med <= max_seconds:  # one real step at the metric's batch, if ~7x the B=1 time still fits (a B=8 step
        # uses the cores better than eight B=1 steps)
        b8 = one(8)
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    per_step = b8 if b8 is not None else 8.0 * med
    return {"value": round(1.0 / per_step, 5), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "scaled": b8 is None, "b1_median_s": round(med, 3), "b8_step_s": round(b8, 3) if b8 is not None else None,
            "sample": f"{len(times)} optimizer steps at B=1 (config 1's batch), SD1.5 UNet + CLIP-L r=4 + KPL, 64x64 latents, fp32 eager "
                      f"oracle; first {skip} discarded, median of the other {len(timed)} = {med:.2f} s/step "
                      f"(all: {', '.join(f'{x:.2f}' for x in times)} s); "
                      + (f"then ONE real step at the metric's batch 8 = {b8:.2f} s, which is `value`" if b8 is not None else
                         "no B=8 step fitted the time budget: `value` = the B=1 median scaled by 8"),
            "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count()}


def make_feeder(step, args, rank):
    """The device feeder of textboost_amd/augment.py on synthetic camera-sized images; returns a callable that fills the step's inputs."""
    import random
    import types

    import numpy as np
    from textboost_amd import augment as aug
    from textboost_amd.data import IndexStream, load_templates

    class WordTokenizer:  # no tokenizer files exist offline: ids from a word hash, padded to 77 like CLIPTokenizer (synthetic, as the ids are)
        model_max_length = 77

        def __call__(self, prompt, truncation=True, padding="max_length", max_length=77, return_tensors="pt"):
            ids = ([49406] + [sum(map(ord, w)) % 49405 for w in prompt.split()])[:max_length - 1]
            return types.SimpleNamespace(input_ids=torch.tensor([ids + [49407] * (max_length - len(ids))], dtype=torch.int64))

    images = []
    for i in range(2):
        r = np.random.default_rng(100 + i)
        yy, xx = np.mgrid[0:1536, 0:2048]
        a = np.clip(np.stack([xx // 8, yy // 6, (xx + yy) // 14], -1) + r.integers(-30, 31, (1536, 2048, 3)), 0, 255).astype(np.uint8)
        images.append((aug.to_device_image(a), ["<sks>"]))
    feeder = aug.DeviceFeeder(images, WordTokenizer(), load_templates("textboost"), size=8 * args.latent, center_crop=False,
                              augment_pipe=aug.PairedAugmentation(hflip="inversion", inversion=True, p=0.8))
    stream = IndexStream(len(images), 42, rank, int(os.environ.get("WORLD_SIZE", "1")))
    random.seed(42 + rank)
    np.random.seed(42 + rank)

    pre = aug.PrefetchFeeder(feeder, args.batch, step.pixel_values, step.input_ids)
    pre.prefetch(stream.take(args.batch))

    def feed():  # called right before each replay: commit the prefetched batch, then produce the next one while the step runs
        pre.commit()
        return lambda: pre.prefetch(stream.take(args.batch))

    return feed


def plan_launch(gpus, env, device_count):
    """What `bench.py --gpus N` does in this process (pure host logic, covered by tests/test_host_logic.py).

    The reference's driver starts one process per GPU with `torchrun --nproc-per-node=len(gpus)` (run_textboost_db.py:106-111) and the
    ranks find each other through RANK / WORLD_SIZE / LOCAL_RANK (train_textboost.py:560, accelerate).  Here:
      * WORLD_SIZE present  -> this process IS one rank of an existing launch (torch.distributed.run): use the environment, and
                               refuse a launch whose size disagrees with --gpus;
      * WORLD_SIZE absent, N == 1 -> the single-GPU run in this process;
      * WORLD_SIZE absent, N > 1  -> this process becomes the launcher: it re-executes bench.py under torch.distributed.run with
                               N ranks on 127.0.0.1 (spawn_ranks) -- after checking that the node has N GPUs."""
    if gpus < 1:
        return {"action": "error", "message": f"--gpus {gpus}: need at least one GPU"}
    if "WORLD_SIZE" in env:
        world, rank, local = int(env["WORLD_SIZE"]), int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
        if world != gpus:
            return {"action": "error", "message": f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} "
                                                  f"(or start bench.py without a launcher: it spawns its own ranks)"}
        if local >= device_count:
            return {"action": "error", "message": f"LOCAL_RANK {local} has no GPU: this node exposes {device_count} device(s); one process "
                                                  f"per GPU is the only supported layout"}
        return {"action": "run", "world": world, "rank": rank, "local": local}
    if gpus > device_count:
        return {"action": "error", "message": f"--gpus {gpus} but this node exposes {device_count} GPU(s) "
                                              f"(torch.cuda.device_count()); one process per GPU is the only supported layout"}
    if gpus == 1:
        return {"action": "run", "world": 1, "rank": 0, "local": 0}
    return {"action": "spawn", "world": gpus}


def launcher_command(gpus, argv, port, python=sys.executable, script=os.path.abspath(__file__)):
    """The torchrun-equivalent command line for N ranks on this node (rendezvous on 127.0.0.1: the hostname may not resolve)."""
    return [python, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script, *argv]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(gpus, argv):
    """Re-execute this script as N ranks; stdout/stderr pass through (rank 0 prints the JSON line last); returns the exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's host driver only does dmabuf IPC (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = launcher_command(gpus, argv, free_port())
    print("bench.py: launching %d ranks: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (BASELINE.json configs[1] is a 250-step run)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE.json metric: 8)")
    ap.add_argument("--latent", type=int, default=None, help="latent side (default 64 = 512^2 images; 96 for --workload sd21)")
    ap.add_argument("--workload", choices=["sd15", "sd21"], default="sd15",
                    help="sd15 = the BASELINE.json metric (configs[1]); sd21 = SURVEY 8(d) config 4 shapes (SD2.x UNet, OpenCLIP-H text "
                         "encoder, LoRA r=8, 96^2 latents) -- a secondary measurement, never the headline")
    ap.add_argument("--vae", action="store_true",
                    help="secondary measurement: the step starts from 8x-larger RGB pixels and runs the SD VAE encoder first "
                         "(train_textboost.py:1036-1037; SURVEY 8(f).1) -- the BASELINE.json metric uses synthetic latents")
    ap.add_argument("--feeder", action="store_true",
                    help="secondary measurement (implies --vae): every step ALSO runs the reference's per-sample dataset work on the device -- "
                         "template draw, PairedAugmentation (p=0.8, inversion), Resize(512, LANCZOS), RandomCrop, normalise, tokenise "
                         "(textboost/dataset.py:353-381; SURVEY 8(f).3) -- from two resident synthetic 1536x2048 instance images")
    ap.add_argument("--fp8-attn", action="store_true",
                    help="BASELINE.json configs[4]: e4m3 P.V in the forward of the hd = 40 self-attention layers (secondary measurement; "
                         "the metric's numerics are fp16)")
    ap.add_argument("--precision", choices=["fp16", "fp32", "bf16"], default="fp16",
                    help="fp16 = --mixed_precision fp16, the reference driver's mode and the BASELINE.json metric; fp32 = the reference's default "
                         "no-AMP mode (README command; exact-fp32 MFMA at 1/16 of the fp16 matrix rate) -- a secondary measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    args.vae = args.vae or args.feeder
    if args.precision == "fp32":
        global MFMA_PEAK_TFLOPS
        MFMA_PEAK_TFLOPS = 157.3  # f32-input MFMA = the fp32 vector rate (MI355X_MICROARCH.md)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product path has no CPU fallback)")
    plan = plan_launch(args.gpus, os.environ, torch.cuda.device_count())
    if plan["action"] == "error":
        raise SystemExit("bench.py: " + plan["message"])
    if plan["action"] == "spawn":  # plain `python bench.py --gpus N`: become the launcher of N ranks (run_textboost_db.py:106-111)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world, rank, local = plan["world"], plan["rank"], plan["local"]
    torch.cuda.set_device(local)
    dist = None
    force_dist = os.environ.get("TB_FORCE_DIST") == "1"  # exercise the RCCL + two-graph path on a single GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world and dist.get_rank() == rank
    from textboost_amd.build import LIB  # noqa: F401
    from textboost_amd import _lib
    _lib.lib()  # fail loudly if the HIP extension is missing
    if os.environ.get("TB_GEMM_VARIANT"):  # kernel-tuning experiments only
        for v in os.environ["TB_GEMM_VARIANT"].split(","):
            _lib.lib().tb_gemm_set_variant(int(v))
    from textboost_amd.workload import build_step

    torch.manual_seed(42)  # the reference seeds every rank identically (train_textboost.py:601); data is offset by rank
    from textboost_amd import models
    sd21 = args.workload == "sd21"
    if args.latent is None:
        args.latent = 96 if sd21 else 64
    step, added = build_step(batch=args.batch, latent=args.latent, data_seed=1000 + rank, world_size=world,
                             device=torch.device("cuda", local), unet_geo=models.SD21_UNET if sd21 else models.SD15_UNET,
                             clip_geo=models.SD21_CLIP if sd21 else models.SD15_CLIP, lora_rank=8 if sd21 else 4, with_vae=args.vae,
                             attn_fp8=args.fp8_attn, precision=args.precision)
    step.force_dist = force_dist
    feed = None
    if args.feeder:
        feed = make_feeder(step, args, rank)
    if args.no_graph:
        for _ in range(2):
            step.step_eager()
    else:
        step.capture(warmup=2)
    def one_step():
        after = feed() if feed is not None else None
        step.replay()
        if after is not None:
            after()

    for _ in range(args.warmup):
        one_step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    sc = step.scalars()
    roof, table = (None, None)
    if rank == 0 and not args.no_roofline:
        roof, table = roofline_leg(step)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not sd21 and not args.vae and args.precision == "fp16":  # the CPU baseline is the headline workload's
        try:
            cpu = cpu_baseline_leg()
        except Exception as e:  # the baseline is a reported reference, never fatal to the GPU number
            cpu = {"error": repr(e)}
    if dist is not None:
        dist.barrier()
    # n_gpus = the ranks the collective library actually connected (not the flag): an N-GPU line can only come from N RCCL ranks
    n_ranks = dist.get_world_size() if (dist is not None and world > 1) else 1
    if rank == 0:
        sps = args.steps * 1.0 / dt
        flop_per_image = FLOP_PER_IMAGE
        metric = "train steps/sec (batch=%d, 512^2, SD1.5, LoRA r=4)" % args.batch  # BASELINE.json's metric at the default batch 8
        workload = ("BASELINE.json configs[1]: SD1.5 UNet (859.5M, frozen, fwd + dgrad bwd) + CLIP-L text encoder LoRA r=4 on "
                    "q/k/v (fwd x2 + bwd x2) + frozen KPL teacher fwd (its rows ride in the student's launches), per-GPU batch %d, %dx%d latents (512^2), 18 added token vectors, "
                    "MSE + 0.1*KPL(cos), GradScaler + clip + AdamW + renorm on device; random-init weights; step as one HIP graph"
                    % (args.batch, args.latent, args.latent))
        if args.vae:
            metric = metric[:-1] + ", + VAE encoder on %d^2 pixels)" % (8 * args.latent)
            workload += "; PLUS the SD VAE encoder (34.2M, fp16 MFMA / fp32 stats) on resident [B,3,%d,%d] fp32 pixels inside the graph" % (
                8 * args.latent, 8 * args.latent)
        if args.feeder:
            metric = metric[:-1] + " + device feeder)"
            workload += ("; PLUS, every step, the dataset work of textboost/dataset.py:353-381 for the batch on the device: PairedAugmentation "
                         "(p=0.8, inversion), Lanczos resize 1536x2048 -> 512, RandomCrop, normalise, prompt tokenisation (word-hash stand-in, cached)")
        if args.fp8_attn:
            metric = metric[:-1] + ", fp8 P.V in the self-attention forward)"
            workload += "; e4m3 P.V (v_mfma_scale_f32_32x32x64_f8f6f4) in the forward of the 64x64-map self-attention layers (BASELINE.json configs[4])"
        if args.precision == "bf16":
            metric = metric[:-1] + ", bf16 mixed precision)"
            workload += "; --mixed_precision bf16: the bfloat16 build of the kernel library (v_mfma_f32_*_bf16), no GradScaler"
        if args.precision == "fp32":
            metric = metric[:-1] + ", fp32 no-AMP mode)"
            workload += "; fp32 (no mixed precision) mode: every weight / activation / gradient fp32, v_mfma_f32_32x32x2_f32, no GradScaler"
        if sd21:  # secondary measurement; algorithmic FLOP taken from the recorded launches of the eager leg
            metric = "train steps/sec (batch=%d, SD2.1 shapes, %d^2 latents, LoRA r=8)" % (args.batch, args.latent)
            workload = ("SURVEY 8(d) config 4: SD2.x UNet (865.9M, Linear proj_in/out, head dim 64) + OpenCLIP-H text encoder (23 layers, "
                        "D=1024) LoRA r=8 + KPL teacher, per-GPU batch %d, %dx%d latents; random-init weights; one HIP graph"
                        % (args.batch, args.latent, args.latent))
            flop_per_image = (roof["recorded_matmul_tflop_per_step"] * 1e12 / args.batch) if roof else float("nan")
        out = {
            # whole-job aggregate: every rank runs a per-GPU-batch step (weak scaling), so the job does `world` of the metric's batch-8 steps per
            # global step; sps is the rate of the slowest rank (the timed region ends at a barrier, dt is the max over ranks)
            "metric": metric, "value": round(sps * world, 4), "unit": "steps/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16"}.get(args.precision, "f16"), "data": "synthetic",
            "config": {"workload": workload,
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "images_per_s": round(sps * args.batch * world, 2)},
            "alg_tflops_per_gpu": round(sps * args.batch * flop_per_image / 1e12, 2),
            "frac_of_mfma_peak_whole_step": round(sps * args.batch * flop_per_image / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "loss": sc["loss"], "loss_scale": sc["loss_scale"], "found_inf_last": sc["found_inf"], "opt_steps": sc["opt_steps"],
            "graph_mode": "eager" if args.no_graph else step.graph_mode,  # "single" | "single+rccl" (all-reduce inside the step graph) | "two+eager-rccl"
            "rccl_world_size": n_ranks if dist is not None else None,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if roof is not None:
            roof["whole_step_frac"] = out["frac_of_mfma_peak_whole_step"]  # 14.1 TFLOP per step x steps/s / 2.5 PFLOP/s
            roof["mfma_busy"] = mfma_busy_table()
            if args.precision == "fp16":
                # `peak` stays the 2.5 PFLOP/s of the microarchitecture guide (the contract); beside it: the rate the same chip holds under pure
                # matrix work, measured now, and the dominant kernel against that
                sp = sustained_mfma_peak()
                roof["sustained_peak"] = sp
                roof["frac_of_sustained_peak"] = round(roof["mfma_frac" if "mfma_frac" in roof else "frac"] * MFMA_PEAK_TFLOPS / sp, 4)
                roof["whole_step_frac_of_sustained_peak"] = round(out["alg_tflops_per_gpu"] / sp, 4)
                roof["dominant_tile_clock"] = dominant_tile_clock(batch=args.batch if args.batch in (1, 2, 4, 8, 16) else 8)
        if table is not None:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernel_table.json"), "w") as f:
                json.dump(table, f, indent=1)
    if dist is not None:
        dist.destroy_process_group()  # before the JSON line: RCCL prints its banner on teardown
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's banner sits in C stdio buffers: flush it so the JSON line is the LAST line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
