"""Full-size (BASELINE.json shapes) parity on the GPU box: SD1.5 / SD2.x UNets and CLIP-L / OpenCLIP-H text encoders with random-init
weights against the fp32 CPU oracle -- B=1..2 forward + backward, the UNet forward at the metric's B=8, the SD2.x UNet at 96x96 latents --
plus size-independent properties of the whole step at B=8.

Tolerances (fp16 MFMA operands / fp32 accumulation vs the fp32 oracle; every check is whole-tensor rel-L2 AND max-abs relative to the
largest reference magnitude AND the worst per-channel rel-L2): UNet prediction 3e-3 / 4e-3 / 4e-3 (measured ~1.1e-3), d(encoder hidden
states) 5e-3 / 6e-3 / 3e-2 (measured ~2e-3; single low-energy channels reach 1.4e-2), encoder hidden states 2e-3 (measured 8e-4),
LoRA / embedding gradients 3e-3 / 4e-3 / 5e-3 (measured ~1e-3).  A faithful fp16 module (oracle/fp16_mode.py) sits 4.9e-3 / 9.5e-3 from
the same fp32 oracle (tests/test_gpu_model.py): the kernels here are inside the reference's own fp16 rounding noise."""
import os
import pytest
import torch

from parity import parity

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def test_sd15_unet_full_size_forward_backward_vs_oracle():
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd import models
    from textboost_amd.unet import HipUNet
    torch.manual_seed(0)
    sd = models.random_state_dict(models.unet_shapes(models.SD15_UNET), 77, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    with torch.device("meta"):
        ref = UNet2DCondition(UNetConfig.sd15())
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    B = 1
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, 64, 64, generator=g).half().float()
    t = torch.tensor([611])
    ehs = torch.randn(B, 77, 768, generator=g).half().float().requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, 64, 64, generator=g)
    pred_ref.backward(dpred)
    hip = HipUNet(models.SD15_UNET, {k: v.to(dev) for k, v in sd.items()}, B, 64, 64, device=dev)
    pred = hip.forward(x.half().to(dev), t.to(dev), ehs.detach().half().view(B * 77, 768).to(dev).contiguous())
    parity("SD1.5 UNet pred", pred, pred_ref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
    d_ehs = hip.backward(dpred.to(dev))
    parity("SD1.5 UNet d_ehs", d_ehs.view(B, 77, 768), ehs.grad, rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=3e-2)
    # BASELINE.json configs[4]: the same UNet with e4m3 P.V in the forward of its five 64x64-map self-attention layers (opt-in).  The
    # attention outputs themselves are 3.6e-2 off on random data (tests/test_gpu_norm_attn.py), but they enter a residual stream that
    # dominates them: on the whole model the mode measures 1.13e-3 (fp16 path 1.09e-3) -- held to 4e-3 / 6e-3 here.
    pred16 = pred.float().clone()
    del hip
    torch.cuda.empty_cache()
    hip8 = HipUNet(models.SD15_UNET, {k: v.to(dev) for k, v in sd.items()}, B, 64, 64, device=dev, attn_fp8=True)
    pred8 = hip8.forward(x.half().to(dev), t.to(dev), ehs.detach().half().view(B * 77, 768).to(dev).contiguous())
    assert not torch.equal(pred8.float(), pred16), "the fp8 attention path did not run"
    parity("SD1.5 UNet pred, fp8 P.V", pred8, pred_ref, rel=4e-3, maxabs=6e-3, ch_dim=1, ch_rel=6e-3)
    parity("SD1.5 UNet d_ehs, fp8 P.V", hip8.backward(dpred.to(dev)).view(B, 77, 768), ehs.grad, rel=6e-3, maxabs=8e-3)


def test_clip_l_full_size_forward_backward_vs_oracle():
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from textboost_amd import models
    from textboost_amd.text_encoder import HipTextEncoder
    torch.manual_seed(0)
    csd = models.random_state_dict(models.clip_shapes(models.SD15_CLIP), 78, device="cpu")
    ref = TextBoostEncoder(CLIPTextCfg.sd15(), r=4)
    ref.load_hf_state_dict(csd)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.02)
        null = ref.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    ref.set_null_embedding(null)
    added = add_tokens(ref, [11, 22, 33])
    B = 2
    hip = HipTextEncoder(models.SD15_CLIP, csd, B, mode="autocast", lora_rank=4, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([11, 22, 33])
    for i, layer in enumerate(ref.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    g = torch.Generator().manual_seed(3)
    ids = ts.synthetic_ids(B, added, g)
    out_ref = ref(ids)
    R = torch.randn(B, 77, 768, generator=g)
    (out_ref * R).sum().backward()
    hip.pack_lora()
    out = hip.forward(ids.to(dev))
    parity("CLIP-L hidden states", out.view(B, 77, 768), out_ref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-2)
    hip.zero_grad()
    hip.backward(R.view(B * 77, 768).to(dev).contiguous())
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in ref.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in ref.layers])
    parity("CLIP-L grad lora_A", hip.grad_A, gA, rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)
    parity("CLIP-L grad lora_B", hip.grad_B, gB, rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)
    parity("CLIP-L grad added rows", hip.grad_added, ref.token_embedding.weight.grad[added], rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)


def test_metric_config_properties_at_batch_8():
    """B=8, 64x64 latents, SD1.5 + CLIP-L (the bench workload): size-independent invariants of the reference step."""
    from textboost_amd import _lib as L
    from textboost_amd.workload import build_step
    torch.manual_seed(42)
    step, added = build_step(batch=8, latent=64)
    te = step.te
    w0 = te.token_table.clone()
    A0, B0 = te.lora_A.clone(), te.lora_B.clone()
    n = 3
    for _ in range(n):
        step.step_eager()
    torch.cuda.synchronize()
    sc = step.scalars()
    assert sc["found_inf"] == 0.0 and sc["opt_steps"] == float(n) and sc["loss_scale"] == 65536.0
    assert 0.5 < sc["loss_mse"] < 2.0 and sc["loss_kpl"] >= 0.0          # eps-prediction of a random UNet vs unit noise
    first = te.first_added
    # rows below min(added_token_ids): gradient zeroed (:1109-1117) -> only AdamW's decoupled decay, every step
    torch.testing.assert_close(te.token_table[:first], w0[:first] * (1 - 1e-3 * 1e-2) ** n, rtol=2e-6, atol=0)
    # added rows moved, stay finite and are norm-clamped to mean_norm (:1138-1149)
    added_rows = te.token_table[first:]
    assert torch.isfinite(added_rows).all() and not torch.equal(added_rows, w0[first:])
    assert (added_rows.norm(dim=-1) <= step.mean_norm * (1 + 1e-5)).all()
    # LoRA: B leaves zero after the first step, A moves after the second (B = 0 makes dA = 0 at step 1: peft gaussian init)
    assert te.lora_B.abs().max() > 0 and not torch.equal(te.lora_A, A0)
    # Adam's per-step move is bounded by ~lr
    assert (te.lora_B - B0).abs().max().item() <= n * 5e-5 * 1.05 + 1e-9
    # pins: null prompts and position 0 come out as the null embedding, bit-exact
    ids = step.ids_all.clone()
    ids[0, 1:] = 49407
    h = te.forward(ids, slot=0).view(ids.shape[0], 77, -1)
    assert torch.equal(h[0], te.null_embedding) and torch.equal(h[3, 0], te.null_embedding[0])


def test_sd21_unet_full_size_forward_backward_vs_oracle():
    """SURVEY 8(d) config 4 shapes: SD2.x UNet (865.9 M; Linear proj_in/out, 5/10/20/20 heads of dim 64, cross dim 1024), B=1, 64^2."""
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd import models
    from textboost_amd.unet import HipUNet
    torch.manual_seed(0)
    assert models.count_params(models.unet_shapes(models.SD21_UNET)) == 865_910_724
    sd = models.random_state_dict(models.unet_shapes(models.SD21_UNET), 79, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    with torch.device("meta"):
        ref = UNet2DCondition(UNetConfig.sd21())
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    B = 1
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 4, 64, 64, generator=g).half().float()
    t = torch.tensor([402])
    ehs = torch.randn(B, 77, 1024, generator=g).half().float().requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, 64, 64, generator=g)
    pred_ref.backward(dpred)
    hip = HipUNet(models.SD21_UNET, {k: v.to(dev) for k, v in sd.items()}, B, 64, 64, device=dev)
    pred = hip.forward(x.half().to(dev), t.to(dev), ehs.detach().half().view(B * 77, 1024).to(dev).contiguous())
    parity("SD2.1 UNet pred", pred, pred_ref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
    d_ehs = hip.backward(dpred.to(dev))
    parity("SD2.1 UNet d_ehs", d_ehs.view(B, 77, 1024), ehs.grad, rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=3e-2)


def test_openclip_h_full_size_forward_backward_vs_oracle():
    """SD2.x text encoder shapes (23 layers, D=1024, 16 heads, erf-GELU MLP), LoRA r=8, B=1."""
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from textboost_amd import models
    from textboost_amd.text_encoder import HipTextEncoder
    torch.manual_seed(0)
    assert models.count_params(models.clip_shapes(models.SD21_CLIP)) == 340_387_840
    csd = models.random_state_dict(models.clip_shapes(models.SD21_CLIP), 80, device="cpu")
    ref = TextBoostEncoder(CLIPTextCfg.sd21(), r=8)
    ref.load_hf_state_dict(csd)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.02)
        null = ref.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    ref.set_null_embedding(null)
    added = add_tokens(ref, [11, 22])
    B = 1
    hip = HipTextEncoder(models.SD21_CLIP, csd, B, mode="autocast", lora_rank=8, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([11, 22])
    for i, layer in enumerate(ref.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    g = torch.Generator().manual_seed(3)
    ids = ts.synthetic_ids(B, added, g)
    out_ref = ref(ids)
    R = torch.randn(B, 77, 1024, generator=g)
    (out_ref * R).sum().backward()
    hip.pack_lora()
    out = hip.forward(ids.to(dev))
    parity("OpenCLIP-H hidden states", out.view(B, 77, 1024), out_ref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-2)
    hip.zero_grad()
    hip.backward(R.view(B * 77, 1024).to(dev).contiguous())
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in ref.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in ref.layers])
    parity("OpenCLIP-H grad lora_A", hip.grad_A, gA, rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)
    parity("OpenCLIP-H grad lora_B", hip.grad_B, gB, rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)
    parity("OpenCLIP-H grad added rows", hip.grad_added, ref.token_embedding.weight.grad[added], rel=3e-3, maxabs=4e-3, ch_dim=0, ch_rel=5e-3)


def _full_unet_pair(geo, cfg, seed, B, hw):
    from oracle.unet_sd import UNet2DCondition
    from textboost_amd import models
    from textboost_amd.unet import HipUNet
    sd = models.random_state_dict(models.unet_shapes(geo), seed, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    with torch.device("meta"):
        ref = UNet2DCondition(cfg)
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    hip = HipUNet(geo, {k: v.to(dev) for k, v in sd.items()}, B, hw, hw, device=dev)
    return ref, hip


# (round 5: the stand-alone SD2.x UNet test at 96x96 latents is covered by test_sd21_full_step_chain_at_96x96_vs_oracle below -- the same launches, pred and
# d_ehs against the same oracle -- and was removed: one full-size 96x96 CPU oracle forward + backward fewer in the `-m gpu` suite)


def test_batch_16_equals_two_batches_of_8():
    """BASELINE.json configs[4] batch (B=16, 64x64 latents, SD1.5): samples are independent, so the UNet forward and its dgrad backward at
    B=16 must reproduce two B=8 runs on the halves -- with M doubled most layers select other tiles / split factors, so this is a parity
    check of those kernels against the B=8 ones that the oracle tests pin.  Both sides are fp16-storage computations with independent
    rounding (each is 1.2e-3 / 1.9e-3 from the fp32 oracle), so they agree to ~sqrt(2) of that, not bit for bit."""
    from textboost_amd import models
    from textboost_amd.unet import HipUNet
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    sd = models.random_state_dict(models.unet_shapes(models.SD15_UNET), 78, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 4, 64, 64, generator=g).half()
    t = torch.randint(0, 1000, (16,), generator=g)
    ehs = torch.randn(16 * 77, 768, generator=g).half()
    dpred = torch.randn(16, 4, 64, 64, generator=g)
    hip16 = HipUNet(models.SD15_UNET, {k: v.to(dev) for k, v in sd.items()}, 16, 64, 64, device=dev)
    pred16 = hip16.forward(x.to(dev), t.to(dev), ehs.to(dev)).float().clone()
    dehs16 = hip16.backward(dpred.to(dev)).float().clone()
    del hip16
    torch.cuda.empty_cache()
    hip8 = HipUNet(models.SD15_UNET, {k: v.to(dev) for k, v in sd.items()}, 8, 64, 64, device=dev)
    for h in range(2):
        sl = slice(8 * h, 8 * h + 8)
        pred8 = hip8.forward(x[sl].to(dev), t[sl].to(dev), ehs[8 * h * 77:(8 * h + 8) * 77].to(dev).contiguous()).float()
        parity(f"B=16 vs B=8 pred, half {h}", pred16[sl], pred8, rel=2.5e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
        dehs8 = hip8.backward(dpred[sl].to(dev)).float()
        parity(f"B=16 vs B=8 d_ehs, half {h}", dehs16[8 * h * 77:(8 * h + 8) * 77], dehs8, rel=4e-3, maxabs=6e-3)


def test_step_invariants_at_batch_16():
    """the whole step at B=16 (configs[4] batch): graph replay == eager bit for bit, finite losses, masked rows only decay."""
    from textboost_amd.workload import build_step
    torch.manual_seed(43)
    step, added = build_step(batch=16, latent=64)
    te = step.te
    w0 = te.token_table.clone()
    step.step_eager()
    torch.cuda.synchronize()
    sc = step.scalars()
    assert sc["found_inf"] == 0.0 and sc["opt_steps"] == 1.0 and 0.5 < sc["loss_mse"] < 2.0
    first = te.first_added
    torch.testing.assert_close(te.token_table[:first], w0[:first] * (1 - 1e-3 * 1e-2), rtol=2e-6, atol=0)
    assert torch.isfinite(te.token_table[first:]).all() and torch.isfinite(te.lora_B).all()


def test_sd15_full_step_at_the_metric_batch_vs_oracle():
    """BASELINE.json configs[1], the benchmarked launches themselves (B=8 tiles, split-K factors, XCD remaps, the hd = 40 attention
    backward) driven by the REAL step object: TWO optimizer steps of `TextBoostStep` (train_textboost.py:1040-1149) at full size -- trainable
    CLIP-L (LoRA r=4, added rows) + the KPL prior / teacher rows -> fp16 hidden states -> SD1.5 UNet -> MSE -> dgrad backward -> d(ehs) ->
    CLIP-L backward -> masks, GradScaler, clip, AdamW (both groups), row decay, renorm -- against the fp32 oracle's `TrainState`.
    Step 1 checks every intermediate (prediction and d(ehs) per sample, losses, gradients); after BOTH steps the updated parameters are compared
    ELEMENTWISE: LoRA A / B, the added token rows, the decay-only rows (north_star: "token-embedding/LoRA weight updates are compared
    elementwise"; tolerances of the small-config test, tests/test_gpu_model.py::test_full_step_matches_oracle_elementwise).
    The oracle evaluates the batch two samples at a time (TrainState.step(chunk=2): samples are independent, the losses are batch means),
    which bounds its memory to a B=2 autograd graph -- TB_TEST_ORACLE_CHUNK to change.
    Then the two KPL teachers at full size (:939, :1096-1100): the default one (teacher rows riding in the student's launches, autocast
    arithmetic) and the reference's plain fp16 module (TB_SEPARATE_TEACHER=1) against the fp32 oracle and against each other, hidden states
    and KPL loss."""
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from oracle.unet_sd import UNetConfig
    from textboost_amd import _lib as L
    from textboost_amd import models, ops
    from textboost_amd.text_encoder import HipTextEncoder
    from textboost_amd.trainer import StepHyper, TextBoostStep
    torch.manual_seed(0)
    B, T, D = 8, 77, 768
    ref_unet, hip_unet = _full_unet_pair(models.SD15_UNET, UNetConfig.sd15(), 83, B, 64)
    csd = models.random_state_dict(models.clip_shapes(models.SD15_CLIP), 84, device="cpu")
    base = TextBoostEncoder(CLIPTextCfg.sd15(), r=0)
    base.load_hf_state_dict(csd)
    ref = TextBoostEncoder(CLIPTextCfg.sd15(), r=4)
    ref.load_hf_state_dict(csd)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.02)                 # (peft starts B at zero: then dA = 0 at step 1 -- non-zero exercises every path twice)
        null = base.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    ref.set_null_embedding(null)
    base.set_null_embedding(null)
    teacher = ts.make_teacher(base)
    added = add_tokens(ref, [11, 22, 33])
    hip = HipTextEncoder(models.SD15_CLIP, csd, B, mode="autocast", lora_rank=4, n_slots=2, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([11, 22, 33])
    for i, layer in enumerate(ref.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    hip_teacher = HipTextEncoder(models.SD15_CLIP, csd, B, mode="half", lora_rank=0, device=dev)
    hip_teacher.set_null_embedding(null)
    st_ref = ts.TrainState(ref, teacher, ref_unet, added, ts.StepConfig())
    step = TextBoostStep(hip_unet, hip, hip_teacher, StepHyper(), (B, 4, 64, 64), device=dev)
    step.external_noise = True
    assert step.merge_teacher and abs(step.mean_norm - st_ref.mean_norm) < 1e-4 * st_ref.mean_norm
    seen = {}
    enc_bwd = step._phase_encoder_backward

    def keep_d_ehs():   # (the encoder backward zeroes the pinned rows of its input in place: copy d(ehs) as the UNet backward delivered it)
        seen["d_ehs"] = step.d_ehs.clone()
        enc_bwd()
    step._phase_encoder_backward = keep_d_ehs
    CH = int(os.environ.get("TB_TEST_ORACLE_CHUNK", "2"))
    g = torch.Generator().manual_seed(6)
    w0 = ref.token_embedding.weight.detach().clone()
    A0 = torch.stack([torch.cat([l.q.lora_A, l.k.lora_A, l.v.lora_A]) for l in ref.layers]).detach().clone()
    B0 = torch.stack([torch.cat([l.q.lora_B, l.k.lora_B, l.v.lora_B]) for l in ref.layers]).detach().clone()
    for it in range(2):
        ids = ts.synthetic_ids(B, added, g)
        pids = ts.synthetic_ids(B, added, g, prior=True)
        if it == 0:
            pids[2, 1:] = 49407                          # one null prior prompt (--null_prob): pinned rows
        x0 = torch.randn(B, 4, 64, 64, generator=g)
        noise = torch.randn(B, 4, 64, 64, generator=g)
        t = torch.tensor([999, 0, 611, 250, 17, 801, 500, 333]) if it == 0 else torch.randint(0, 1000, (B,), generator=g)
        out = st_ref.step(x0, noise, t, ids, pids, chunk=CH)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        torch.cuda.synchronize()
        sc = step.scalars()
        assert sc["found_inf"] == 0.0 and sc["opt_steps"] == float(it + 1) and sc["loss_scale"] == 65536.0, sc
        print(f"[parity] B=8 step {it}: mse {sc['loss_mse']:.6f} (oracle {out['mse']:.6f})  kpl {sc['loss_kpl']:.6e} (oracle {out['kpl']:.6e})  "
              f"grad norm {sc['grad_norm']:.5e} (oracle {out['lora_grad_norm']:.5e})")
        assert abs(sc["loss_mse"] - out["mse"]) < 1e-2 * abs(out["mse"]), (sc, out["mse"])
        assert abs(sc["loss_kpl"] - out["kpl"]) < 5e-2 * abs(out["kpl"]) + 1e-5, (sc, out["kpl"])
        assert abs(sc["grad_norm"] - out["lora_grad_norm"]) < 2e-2 * out["lora_grad_norm"]
        inv = 1.0 / 65536.0
        if it == 0:   # every intermediate of the differentiable chain, no sample hiding behind the others (each has its own timestep)
            parity("B=8 step: UNet pred", step.pred, out["pred"], rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
            d_ehs = seen["d_ehs"].view(B, T, D) * inv
            parity("B=8 step: d_ehs", d_ehs, out["d_ehs"], rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=3e-2)
            for b in range(B):
                parity(f"  pred sample {b} (t={int(t[b])})", step.pred[b], out["pred"][b], rel=3e-3, maxabs=5e-3, verbose=False)
                parity(f"  d_ehs sample {b}", d_ehs[b], out["d_ehs"][b], rel=6e-3, maxabs=8e-3, verbose=False)
            nb = step.ids_all.shape[0]
            assert torch.equal(step.h_teacher.view(B, T, D)[2].cpu(), null)       # the null prior prompt of the teacher rows: pinned, bit-exact
        # gradients (the flat buffer holds loss_scale * grad; the oracle's LoRA gradients are post-clip)
        nl = len(ref.layers)
        gA = torch.stack([torch.cat(out["g_lora"][6 * l + 0: 6 * l + 6: 2]) for l in range(nl)])
        gB = torch.stack([torch.cat(out["g_lora"][6 * l + 1: 6 * l + 6: 2]) for l in range(nl)])
        clip = min(1.0, 1.0 / (out["lora_grad_norm"] + 1e-6))
        parity(f"B=8 step {it}: grad lora_A", step.te.grad_A * inv * clip, gA, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
        parity(f"B=8 step {it}: grad lora_B", step.te.grad_B * inv * clip, gB, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
        parity(f"B=8 step {it}: grad added rows", step.te.grad_added * inv, out["g_emb_added"], rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
        # ---- the parameters after the update, ELEMENTWISE (the optimizer tail :1128-1149 at full size)
        w, wr = step.te.token_table.cpu(), ref.token_embedding.weight.detach()
        torch.testing.assert_close(w[:49408], wr[:49408], rtol=1e-6, atol=1e-7)                       # decay-only rows
        # added rows: Adam's first moves are ~emb_lr * sign(g), so an element whose gradient sits inside the fp16 path's noise may move the other way
        # (|difference| = 2 * emb_lr; one such element of the 3 x 768 is already a whole-tensor rel-L2 of 2.1e-3 here).  Bound the NUMBER of such
        # elements and hold everything else tight (measured: 0 or 1 flipped element, the others within 2.5e-5)
        dE_abs = (w[added] - wr[added]).abs()
        assert dE_abs.max().item() < 2.5e-3                                                           # <= ~2 * emb_lr
        flipped = dE_abs > 1e-3
        assert int(flipped.sum()) <= max(2, int(0.002 * flipped.numel())), int(flipped.sum())
        assert rel_err(torch.where(flipped, wr[added], w[added]), wr[added]) < 5e-4
        A_ref = torch.stack([torch.cat([l.q.lora_A, l.k.lora_A, l.v.lora_A]) for l in ref.layers]).detach()
        B_ref = torch.stack([torch.cat([l.q.lora_B, l.k.lora_B, l.v.lora_B]) for l in ref.layers]).detach()
        A_hip, B_hip = step.te.lora_A.cpu(), step.te.lora_B.cpu()
        assert (A_hip - A_ref).abs().max().item() < 1.5e-4 and (B_hip - B_ref).abs().max().item() < 1.5e-4      # <= ~2 * lr
        # the MOVES themselves: Adam's first steps are ~lr * sign(g), so an element whose gradient is inside the fp16 path's noise can flip --
        # the update vectors as a whole must still point the oracle's way (measured: see the printed values)
        dA, dB, dE = rel_err(A_hip - A0, A_ref - A0), rel_err(B_hip - B0, B_ref - B0), rel_err(w[added] - w0[added], wr[added] - w0[added])
        print(f"[parity] B=8 after step {it}: rel-L2 of the parameter MOVES  lora_A {dA:.3e}  lora_B {dB:.3e}  added rows {dE:.3e};  "
              f"max |param - oracle| lora_A {(A_hip - A_ref).abs().max().item():.2e} lora_B {(B_hip - B_ref).abs().max().item():.2e} "
              f"added {(w[added] - wr[added]).abs().max().item():.2e}")
        assert dA < 8e-2 and dB < 8e-2 and dE < 8e-2      # measured 2.5e-2 / 3.9e-2 / 4.8e-4 (3.5e-2 with one flipped element) after step 1, 1.4e-2 / 2.1e-2 / 2.6e-3 after step 2
    assert step.scalars()["opt_steps"] == 2.0
    # ---- the two teachers (frozen encoder on the prior prompts, :939 / :1096-1100) at full size
    pids_d = step.prior_ids
    with torch.no_grad():
        t_ref = teacher(pids.clone())
        h_ref = ref(pids.clone()).float()
        kpl_ref = (1 - torch.nn.functional.cosine_similarity(h_ref, t_ref.float(), dim=-1)).mean().item()
    hip.pack_lora()
    out_m = hip.forward(step.ids_all, slot=0, extra_ids=pids_d, extra_table=step.teacher_table32)
    nb = step.ids_all.shape[0]
    h_prior, h_merged = out_m[B * T:nb * T], out_m[nb * T:]
    h_sep = hip_teacher.forward(pids_d, slot=0)
    parity("teacher rows inside the student pass (default) vs fp32 oracle", h_merged.view(B, T, D), t_ref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-2)
    parity("separate fp16 teacher module (TB_SEPARATE_TEACHER=1, the reference's :939) vs fp32 oracle", h_sep.view(B, T, D), t_ref, rel=3e-3,
           maxabs=6e-3)                                  # measured 1.3e-3 / 2.4e-3 (the merged rows: 7.8e-4 / 1.0e-3)
    parity("merged teacher vs separate fp16 teacher", h_merged.view(B, T, D), h_sep.view(B, T, D), rel=3e-3, maxabs=6e-3)   # measured 1.2e-3 / 2.6e-3
    kl = []
    for h0 in (h_merged, h_sep):
        stt = torch.zeros(L.ST_COUNT, device=dev)
        stt[L.ST_LOSS_SCALE] = 1.0
        ops.kpl_cos(h_prior, h0, torch.empty(B * T, D, device=dev), torch.empty(B * T, device=dev), stt[L.ST_LOSS_KPL:], stt[L.ST_LOSS_SCALE:], 0.1)
        kl.append(stt[L.ST_LOSS_KPL].item())
    print(f"[parity] KPL loss at full size: merged teacher {kl[0]:.6e}, separate fp16 teacher {kl[1]:.6e}, fp32 oracle {kpl_ref:.6e}")
    # measured: 2.528434e-01 / 2.528739e-01 / 2.528545e-01 -- the benchmarked (merged) teacher is the CLOSER one to the fp32 oracle
    assert abs(kl[0] - kpl_ref) < 1e-3 * kpl_ref and abs(kl[1] - kpl_ref) < 2e-3 * kpl_ref and abs(kl[0] - kl[1]) < 2e-3 * kpl_ref


def test_sd21_full_step_chain_at_96x96_vs_oracle():
    """BASELINE.json configs[3] as ONE chain (the pieces above test its halves): trainable OpenCLIP-H (23 layers, LoRA r=8, added rows) -> fp16
    hidden states -> SD2.x UNet at 768^2 images = 96x96 latents -> dgrad backward -> d(ehs) -> encoder backward -> LoRA A / B and added-row
    gradients (train_textboost.py:1054-1067, :1108), B=2 with the oracle one sample at a time; v-prediction only changes the target of the MSE
    (tested on the small config), not this chain.  Tolerances as for the SD1.5 chain at the metric batch."""
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from oracle.unet_sd import UNetConfig
    from textboost_amd import models, ops
    from textboost_amd.text_encoder import HipTextEncoder
    torch.manual_seed(0)
    B, T, D, hw = 2, 77, 1024, 96
    ref_unet, hip_unet = _full_unet_pair(models.SD21_UNET, UNetConfig.sd21(), 85, B, hw)
    for p in ref_unet.parameters():
        p.requires_grad_(False)
    csd = models.random_state_dict(models.clip_shapes(models.SD21_CLIP), 86, device="cpu")
    ref = TextBoostEncoder(CLIPTextCfg.sd21(), r=8)
    ref.load_hf_state_dict(csd)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.02)
        null = ref.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    ref.set_null_embedding(null)
    added = add_tokens(ref, [11, 22, 33])
    hip = HipTextEncoder(models.SD21_CLIP, csd, B, mode="autocast", lora_rank=8, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([11, 22, 33])
    for i, layer in enumerate(ref.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    g = torch.Generator().manual_seed(7)
    ids = ts.synthetic_ids(B, added, g)
    x = torch.randn(B, 4, hw, hw, generator=g).half().float()
    t = torch.tensor([731, 48])
    dpred = torch.randn(B, 4, hw, hw, generator=g)
    preds, dehs = [], []
    for b in range(B):
        h = ref(ids[b:b + 1])
        h.retain_grad()
        p = ref_unet(x[b:b + 1], t[b:b + 1], h)
        (p * dpred[b:b + 1]).sum().backward()
        preds.append(p.detach())
        dehs.append(h.grad.detach())
    pred_ref, dehs_ref = torch.cat(preds), torch.cat(dehs)
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in ref.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in ref.layers])
    gE = ref.token_embedding.weight.grad[added]
    hip.pack_lora()
    h_hip = hip.forward(ids.to(dev))
    ehs16 = torch.empty(B * T, D, device=dev, dtype=torch.float16)
    ops.convert(h_hip, ehs16)
    pred = hip_unet.forward(x.half().to(dev), t.to(dev), ehs16)
    parity("SD2.1 chain: UNet pred @96x96", pred, pred_ref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
    d_ehs = hip_unet.backward(dpred.to(dev))
    # (worst of 1024 hidden channels over only 2 x 77 tokens: 3.1e-2 measured on a channel whose gradient is ~30x below the median one)
    parity("SD2.1 chain: d_ehs", d_ehs.view(B, T, D), dehs_ref, rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=4e-2)
    hip.zero_grad()
    hip.backward(d_ehs.float().contiguous())
    parity("SD2.1 chain: grad lora_A", hip.grad_A, gA, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
    parity("SD2.1 chain: grad lora_B", hip.grad_B, gB, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
    parity("SD2.1 chain: grad added rows", hip.grad_added, gE, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
