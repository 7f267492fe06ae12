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
    backward): ONE full-size forward + backward of the step's differentiable chain -- trainable CLIP-L (LoRA r=4, added rows) -> fp16
    hidden states -> SD1.5 UNet forward -> UNet dgrad backward -> d(ehs) -> CLIP-L backward -> LoRA A / B and added-row gradients
    (train_textboost.py:1054-1067, :1108) -- against the fp32 oracle, plus the frozen KPL teacher rows that ride in the student's launches
    (:1096-1100).  The oracle runs the 8 samples one at a time (samples are independent; parameter gradients accumulate), which bounds
    its memory to a B=1 autograd graph.  Tolerances: the existing full-size ones for pred / d_ehs; gradients that went through BOTH
    networks 4e-3 / 6e-3 / 1e-2 (the small-config whole-step bound)."""
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from oracle.unet_sd import UNetConfig
    from textboost_amd import models
    from textboost_amd.text_encoder import HipTextEncoder
    torch.manual_seed(0)
    B, T, D = 8, 77, 768
    ref_unet, hip_unet = _full_unet_pair(models.SD15_UNET, UNetConfig.sd15(), 83, B, 64)
    for p in ref_unet.parameters():
        p.requires_grad_(False)                      # frozen (:696): dgrad only, like the product
    csd = models.random_state_dict(models.clip_shapes(models.SD15_CLIP), 84, device="cpu")
    base = TextBoostEncoder(CLIPTextCfg.sd15(), r=0)
    base.load_hf_state_dict(csd)
    ref = TextBoostEncoder(CLIPTextCfg.sd15(), r=4)
    ref.load_hf_state_dict(csd)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.02)
        null = base.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    ref.set_null_embedding(null)
    base.set_null_embedding(null)
    added = add_tokens(ref, [11, 22, 33])
    hip = HipTextEncoder(models.SD15_CLIP, csd, B, mode="autocast", lora_rank=4, n_slots=2, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([11, 22, 33])
    for i, layer in enumerate(ref.layers):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    g = torch.Generator().manual_seed(6)
    ids = ts.synthetic_ids(B, added, g)
    pids = ts.synthetic_ids(B, added, g, prior=True)
    pids[2, 1:] = 49407                              # one null prior prompt (--null_prob): pinned rows
    x = torch.randn(B, 4, 64, 64, generator=g).half().float()
    t = torch.tensor([999, 0, 611, 250, 17, 801, 500, 333])
    dpred = torch.randn(B, 4, 64, 64, generator=g)
    # ---- oracle, a few samples at a time (independent samples, accumulated parameter gradients; 2 per pass = two B=1 autograd graphs of host
    # memory and ~25 % less CPU time than one at a time -- TB_TEST_ORACLE_CHUNK to change)
    preds, dehs = [], []
    CH = int(os.environ.get("TB_TEST_ORACLE_CHUNK", "2"))
    for b in range(0, B, CH):
        h = ref(ids[b:b + CH])
        h.retain_grad()
        p = ref_unet(x[b:b + CH], t[b:b + CH], h)
        (p * dpred[b:b + CH]).sum().backward()
        preds.append(p.detach())
        dehs.append(h.grad.detach())
    pred_ref, dehs_ref = torch.cat(preds), torch.cat(dehs)
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in ref.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in ref.layers])
    gE = ref.token_embedding.weight.grad[added]
    with torch.no_grad():
        teacher_ref = base(pids)
    # ---- HIP path: the step's own call sequence (trainer._phase_student / _phase_unet_* / _phase_encoder_backward)
    from textboost_amd import ops
    hip.pack_lora()
    table0 = torch.empty(49408, D)
    table0.copy_(csd["text_model.embeddings.token_embedding.weight"])
    out = hip.forward(ids.to(dev), slot=0, extra_ids=pids.to(dev), extra_table=table0.to(dev))
    h_hip, h_teacher = out[:B * T], out[B * T:]
    parity("B=8 teacher rows inside the student pass (CLIP-L)", h_teacher.view(B, T, D), teacher_ref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-2)
    assert torch.equal(h_teacher.view(B, T, D)[2].cpu(), null)
    ehs16 = torch.empty(B * T, D, device=dev, dtype=torch.float16)
    ops.convert(h_hip, ehs16)
    pred = hip_unet.forward(x.half().to(dev), t.to(dev), ehs16)
    parity("B=8 step: UNet pred", pred, pred_ref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
    for b in range(B):  # no sample may hide behind the others (every sample has its own timestep); this is also the round-2 "UNet forward at the
        # metric batch" check -- same launches, folded into this test so that the suite builds ONE full-size B=8 CPU oracle pass, not two
        parity(f"  pred sample {b} (t={int(t[b])})", pred[b], pred_ref[b], rel=3e-3, maxabs=5e-3, verbose=False)
    d_ehs = hip_unet.backward(dpred.to(dev))
    parity("B=8 step: d_ehs", d_ehs.view(B, T, D), dehs_ref, rel=5e-3, maxabs=6e-3, ch_dim=2, ch_rel=3e-2)
    for b in range(B):  # no sample may hide behind the others
        parity(f"  d_ehs sample {b}", d_ehs.view(B, T, D)[b], dehs_ref[b], rel=6e-3, maxabs=8e-3, verbose=False)
    hip.zero_grad()
    hip.backward(d_ehs.float().contiguous(), slot=0)
    parity("B=8 step: grad lora_A", hip.grad_A, gA, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
    parity("B=8 step: grad lora_B", hip.grad_B, gB, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)
    parity("B=8 step: grad added rows", hip.grad_added, gE, rel=4e-3, maxabs=6e-3, ch_dim=0, ch_rel=1e-2)


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
