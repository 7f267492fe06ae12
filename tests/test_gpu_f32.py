"""fp32 (no-AMP) numeric mode, kernel level: every fp32 entry point of csrc/f32_path.hip against float64 torch on the CPU.

The reference trains in fp32 unless --mixed_precision fp16 is passed (train_textboost.py:298-308, :930-939; README.md:58-76).  Contractions
run on the exact-fp32 matrix instruction, so the only difference from an fp32 CPU run is summation order: tolerance 2e-6 rel-L2 /
1e-5 max-abs (relative to the largest reference magnitude) for the GEMM family, 1e-5 / 5e-5 where exp / rsqrt are involved."""
import pytest
import torch
import torch.nn.functional as F

from parity import parity
from test_gpu_gemm import rel_err

pytestmark = pytest.mark.gpu
dev = "cuda"


def _ops():
    from textboost_amd import ops
    return ops


def _L():
    from textboost_amd import _lib as L
    return L


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (100, 70, 52), (1, 5, 4), (300, 129, 200), (77, 768, 768)])
def test_gemm_f32_linear_epilogues(M, N, K):
    ops, L = _ops(), _L()
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K + 8, generator=g)[:, 3:3 + K].to(dev)        # unaligned column slice: the scalar load path
    W = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    R = torch.randn(M, N, generator=g).to(dev)
    ref = A.double().cpu() @ W.double().cpu().t()
    out = torch.empty(M, N, device=dev)
    ops.gemm(A, W, out)
    parity("gemm_f32", out, ref, rel=2e-6, maxabs=1e-5)
    ops.gemm(A, W, out, bias=bias, R=R, alpha=0.5)
    parity("gemm_f32 + bias + R", out, 0.5 * ref + bias.double().cpu() + R.double().cpu(), rel=2e-6, maxabs=1e-5)
    pre = torch.empty(M, N, device=dev)
    for act, fn in ((L.ACT_QUICK_GELU, lambda x: x * torch.sigmoid(1.702 * x)), (L.ACT_GELU, lambda x: F.gelu(x)), (L.ACT_SILU, F.silu)):
        ops.gemm(A, W, out, bias=bias, act=act, C2=pre if act != L.ACT_SILU else None)
        z = ref + bias.double().cpu()
        parity(f"gemm_f32 act {act}", out, fn(z), rel=1e-5, maxabs=5e-5)
        if act != L.ACT_SILU:
            parity("  saved pre-activation", pre, z, rel=2e-6, maxabs=1e-5)
    # activation-gradient epilogues: v * act'(C2)
    z = torch.randn(M, N, generator=g).to(dev)
    zc = z.double().cpu().requires_grad_(True)
    (zc * torch.sigmoid(1.702 * zc)).sum().backward()
    ops.gemm(A, W, out, act=L.ACT_QUICK_GELU_GRAD, C2=z)
    parity("gemm_f32 quick-gelu grad", out, ref * zc.grad, rel=1e-5, maxabs=5e-5)
    zc.grad = None
    F.gelu(zc).sum().backward()
    ops.gemm(A, W, out, act=L.ACT_GELU_GRAD, C2=z)
    parity("gemm_f32 gelu grad", out, ref * zc.grad, rel=1e-5, maxabs=5e-5)


def test_gemm_f32_k_extension_rowbias_and_transposed_views():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    M, N, K1, K2 = 154, 96, 64, 16
    A, A2 = torch.randn(M, K1, generator=g).to(dev), torch.randn(M, K2, generator=g).to(dev)
    W, W2 = torch.randn(N, K1, generator=g).to(dev), torch.randn(N, K2, generator=g).to(dev)
    rb = torch.randn(2, N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    ops.gemm(A, W, out, A2=A2, W2=W2, rowbias=rb, rows_per_group=77)
    ref = A.double().cpu() @ W.double().cpu().t() + A2.double().cpu() @ W2.double().cpu().t() + rb.double().cpu().repeat_interleave(77, 0)
    parity("gemm_f32 K-extension + rowbias", out, ref, rel=2e-6, maxabs=1e-5)
    # transposed operand views (LoRA weight gradients): out[M,N] = At^T @ Wt, accumulated onto R
    Mt, Nt, Kt = 48, 12, 231
    At, Wt = torch.randn(Kt, Mt, generator=g).to(dev), torch.randn(Kt, Nt, generator=g).to(dev)
    acc = torch.randn(Mt, Nt, generator=g).to(dev)
    ref = acc.double().cpu() + 0.25 * At.double().cpu().t() @ Wt.double().cpu()
    ops.gemm_f32_t(At, Wt, acc, Mt, Nt, Kt, a_trans=True, w_trans=True, R=acc, alpha=0.25)
    parity("gemm_f32_t", acc, ref, rel=2e-6, maxabs=1e-5)


def _nhwc(x):  # [B,C,H,W] -> [B*H*W, C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


@pytest.mark.parametrize("mode", ["s1", "dgrad", "s2", "up", "tr"])
def test_gemm_f32_conv_gathers(mode):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    B, Ci, Co, H, W = 2, 8, 12, 10, 6
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g)
    wf = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(dev)      # [Co][tap][Ci]
    wd = w.permute(1, 2, 3, 0).reshape(Ci, 9 * Co).contiguous().to(dev)      # dgrad operand [Ci][tap][Co]
    if mode == "s1":
        ref = F.conv2d(x.double(), w.double(), padding=1)
        out = torch.empty(B * H * W, Co, device=dev)
        ops.gemm(_nhwc(x).to(dev), wf, out, conv=dict(B=B, Hin=H, Win=W, Cin=Ci, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=0))
    elif mode == "dgrad":     # input gradient of the stride-1 conv: taps mirrored (sign = -1) on the transposed weights
        dy = torch.randn(B, Co, H, W, generator=g)
        xr = x.double().requires_grad_(True)
        F.conv2d(xr, w.double(), padding=1).backward(dy.double())
        ref = xr.grad
        out = torch.empty(B * H * W, Ci, device=dev)
        ops.gemm(_nhwc(dy).to(dev), wd, out, conv=dict(B=B, Hin=H, Win=W, Cin=Co, Hout=H, Wout=W, stride=1, sign=-1, upsample=0, transposed=0))
    elif mode == "s2":
        ref = F.conv2d(x.double(), w.double(), padding=1, stride=2)
        Ho, Wo = ref.shape[2:]
        out = torch.empty(B * Ho * Wo, Co, device=dev)
        ops.gemm(_nhwc(x).to(dev), wf, out, conv=dict(B=B, Hin=H, Win=W, Cin=Ci, Hout=Ho, Wout=Wo, stride=2, sign=1, upsample=0, transposed=0))
    elif mode == "up":
        ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), padding=1)
        out = torch.empty(B * 4 * H * W, Co, device=dev)
        ops.gemm(_nhwc(x).to(dev), wf, out, conv=dict(B=B, Hin=H, Win=W, Cin=Ci, Hout=2 * H, Wout=2 * W, stride=1, sign=1, upsample=1, transposed=0))
    else:                     # input gradient of the stride-2 conv (transposed gather)
        xr = x.double().requires_grad_(True)
        y = F.conv2d(xr, w.double(), padding=1, stride=2)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy.double())
        ref = xr.grad
        out = torch.empty(B * H * W, Ci, device=dev)
        ops.gemm(_nhwc(dy).to(dev), wd, out, conv=dict(B=B, Hin=dy.shape[2], Win=dy.shape[3], Cin=Co, Hout=H, Wout=W, stride=2, sign=1,
                                                       upsample=0, transposed=1))
    parity(f"gemm_f32 conv {mode}", out, _nhwc(ref), rel=2e-6, maxabs=1e-5)


def test_gemm_f32_geglu_forward_and_backward_epilogues():
    from textboost_amd.unet import pack_geglu_rows
    ops, L = _ops(), _L()
    g = torch.Generator().manual_seed(3)
    M, C = 90, 64
    x = torch.randn(M, C, generator=g)
    w = torch.randn(8 * C, C, generator=g) * 0.2
    b = torch.randn(8 * C, generator=g)
    w2 = torch.randn(C, 4 * C, generator=g) * 0.1
    xr = x.double().requires_grad_(True)
    proj = xr @ w.double().t() + b.double()
    h, gate = proj.chunk(2, dim=-1)
    gated = h * F.gelu(gate)
    dy = torch.randn(M, C, generator=g)
    out_ref = gated @ w2.double().t()
    out_ref.backward(dy.double())
    wp, bp = pack_geglu_rows(w).to(dev), pack_geglu_rows(b).to(dev)
    raw = torch.empty(M, 8 * C, device=dev)
    got = torch.empty(M, 4 * C, device=dev)
    ops.gemm(x.to(dev), wp, got, bias=bp, act=L.ACT_GEGLU, C2=raw)
    parity("gemm_f32 GEGLU", got, gated, rel=1e-5, maxabs=5e-5)
    # backward: d(gated) = dy @ w2 in the ff.net.2 dgrad GEMM, GEGLU backward in its epilogue -> packed d(proj); then the ff1 dgrad
    dproj = torch.empty(M, 8 * C, device=dev)
    ops.gemm(dy.to(dev), w2.t().contiguous().to(dev), dproj, act=L.ACT_GEGLU_GRAD, C2=raw)
    dx = torch.empty(M, C, device=dev)
    ops.gemm(dproj, wp.t().contiguous(), dx)
    parity("gemm_f32 GEGLU backward -> dx", dx, xr.grad, rel=1e-5, maxabs=5e-5)


@pytest.mark.parametrize("B,HW,C,silu", [(2, 64, 64, True), (3, 100, 320, False), (1, 16, 1280, True)])
def test_groupnorm_f32(B, HW, C, silu):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    xb = torch.randn(B * HW, C + 8, generator=g) * 2 + 0.5
    x = xb[:, 4:4 + C].to(dev)
    gamma, beta = (torch.randn(C, generator=g) * 0.5 + 1).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    xr = x.double().cpu().view(B, HW, C).permute(0, 2, 1).requires_grad_(True)
    y = F.group_norm(xr, 32, gamma.double().cpu(), beta.double().cpu(), 1e-5)
    if silu:
        y = F.silu(y)
    dy = torch.randn(B * HW, C, generator=g)
    add = torch.randn(B * HW, C, generator=g)
    y.backward(dy.double().view(B, HW, C).permute(0, 2, 1))
    out, stats = torch.empty(B * HW, C, device=dev), torch.empty(B * 32, 2, device=dev)
    ops.groupnorm_fwd(x, out, gamma, beta, stats, None, B, HW, C, 32, 1e-5, silu)
    parity("groupnorm_f32 fwd", out, y.permute(0, 2, 1).reshape(B * HW, C), rel=1e-5, maxabs=5e-5)
    dx = torch.empty(B * HW, C, device=dev)
    ops.groupnorm_bwd(dy.to(dev), x, gamma, beta, stats, dx, None, B, HW, C, 32, silu, add=add.to(dev))
    parity("groupnorm_f32 bwd", dx, xr.grad.permute(0, 2, 1).reshape(B * HW, C) + add.double(), rel=1e-5, maxabs=5e-5)


@pytest.mark.parametrize("B,H,Sq,Skv,hd,causal", [(2, 4, 96, 96, 40, False), (1, 8, 200, 77, 40, False), (2, 3, 77, 77, 64, True),
                                                  (1, 2, 130, 70, 160, False)])
def test_attention_f32(B, H, Sq, Skv, hd, causal):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    C = H * hd
    q = torch.randn(B * Sq, C, generator=g).to(dev)
    kv = torch.randn(B * Skv, 2 * C, generator=g).to(dev)
    k, v = kv[:, :C], kv[:, C:]
    qr, kr, vr = [t.double().cpu().reshape(B, -1, H, hd).transpose(1, 2).requires_grad_(True) for t in (q, k, v)]
    s = qr @ kr.transpose(-1, -2) * hd ** -0.5
    if causal:
        s = s + torch.full((Sq, Skv), float("-inf"), dtype=torch.float64).triu(1)
    oref = (torch.softmax(s, -1) @ vr).transpose(1, 2).reshape(B * Sq, C)
    o, lse = torch.empty(B * Sq, C, device=dev), torch.empty(B, H, Sq, device=dev)
    ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd, causal=causal)
    parity("attention_f32 O", o, oref, rel=1e-5, maxabs=5e-5)
    parity("attention_f32 LSE", lse, torch.logsumexp(s, -1), rel=1e-5, maxabs=5e-5)
    do = torch.randn(B * Sq, C, generator=g)
    oref.backward(do.double())
    delta = torch.empty(B, H, Sq, device=dev)
    dq, dkv = torch.empty(B * Sq, C, device=dev), torch.empty(B * Skv, 2 * C, device=dev)
    ops.attention_bwd(q, k, v, o, lse, do.to(dev), delta, dq, dkv[:, :C], dkv[:, C:], B, H, Sq, Skv, hd, causal=causal)
    back = lambda t, S: t.transpose(1, 2).reshape(B * S, C)  # noqa: E731
    parity("attention_f32 dQ", dq, back(qr.grad, Sq), rel=1e-5, maxabs=5e-5)
    parity("attention_f32 dK", dkv[:, :C], back(kr.grad, Skv), rel=1e-5, maxabs=5e-5)
    parity("attention_f32 dV", dkv[:, C:], back(vr.grad, Skv), rel=1e-5, maxabs=5e-5)


def test_streaming_kernels_f32():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    B, H, W, C = 2, 6, 4, 16
    # add_noise / velocity
    from textboost_amd.trainer import alphas_cumprod
    acp = alphas_cumprod(device=dev)
    x0, noise = torch.randn(B, 4, H, W, generator=g).to(dev), torch.randn(B, 4, H, W, generator=g).to(dev)
    t = torch.tensor([999, 3], device=dev)
    noisy, vel = torch.empty_like(x0), torch.empty_like(x0)
    ops.add_noise(x0, noise, t, acp, noisy, vel)
    a = acp[t].double().cpu().view(B, 1, 1, 1)
    parity("add_noise_f32", noisy, a.sqrt() * x0.double().cpu() + (1 - a).sqrt() * noise.double().cpu(), rel=1e-6, maxabs=1e-6)
    parity("velocity_f32", vel, a.sqrt() * noise.double().cpu() - (1 - a).sqrt() * x0.double().cpu(), rel=1e-6, maxabs=1e-6)
    # timestep embedding [cos | sin]
    te = torch.empty(B, 320, device=dev)
    ops.timestep_embed(t, te)
    k = torch.arange(160, dtype=torch.float64)
    arg = t.double().cpu()[:, None] * torch.exp(-9.210340371976184 * k / 160)
    parity("timestep_embed_f32", te, torch.cat([arg.cos(), arg.sin()], 1), rel=2e-5, maxabs=1e-4)   # fp32 argument t * f up to 999
    # boundary convs
    xin = torch.randn(B, 4, H, W, generator=g)
    w = torch.randn(C, 4, 3, 3, generator=g)
    bias = torch.randn(C, generator=g)
    out = torch.empty(B * H * W, C, device=dev)
    ops.conv4_to_nhwc(xin.to(dev), w.permute(2, 3, 1, 0).reshape(36, C).contiguous().to(dev), bias.to(dev), out, B, H, W, C, sign=1)
    parity("conv4_to_nhwc_f32", out, _nhwc(F.conv2d(xin.double(), w.double(), bias.double(), padding=1)), rel=2e-6, maxabs=1e-5)
    wo = torch.randn(4, C, 3, 3, generator=g)
    bo = torch.randn(4, generator=g)
    a_nhwc = torch.randn(B * H * W, C, generator=g)
    pred = torch.empty(B, 4, H, W, device=dev)
    ops.conv_to4(a_nhwc.to(dev), wo.permute(0, 2, 3, 1).reshape(4, 9, C).contiguous().to(dev), bo.to(dev), pred, B, H, W, C)
    a_nchw = a_nhwc.view(B, H, W, C).permute(0, 3, 1, 2)
    parity("conv_to4_f32", pred, F.conv2d(a_nchw.double(), wo.double(), bo.double(), padding=1), rel=2e-6, maxabs=1e-5)
    # conv_out input gradient through conv4_to_nhwc(sign=-1)
    dpred = torch.randn(B, 4, H, W, generator=g)
    ar = a_nchw.double().requires_grad_(True)
    F.conv2d(ar, wo.double(), padding=1).backward(dpred.double())
    da = torch.empty(B * H * W, C, device=dev)
    ops.conv4_to_nhwc(dpred.to(dev), wo.permute(2, 3, 0, 1).reshape(36, C).contiguous().to(dev), None, da, B, H, W, C, sign=-1)
    parity("conv_out dgrad f32", da, _nhwc(ar.grad), rel=2e-6, maxabs=1e-5)
    # upsample / its backward / add / mse
    xs = torch.randn(B * H * W, C, generator=g).to(dev)
    u = torch.empty(B * 4 * H * W, C, device=dev)
    ops.upsample2x(xs, u, B, H, W, C)
    uref = F.interpolate(xs.cpu().view(B, H, W, C).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    assert torch.equal(u.cpu(), _nhwc(uref))
    du = torch.randn(B * 4 * H * W, C, generator=g).to(dev)
    dx = torch.empty(B * H * W, C, device=dev)
    ops.pool2x2_sum(du, dx, B, H, W, C)
    dref = F.avg_pool2d(du.double().cpu().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2), 2) * 4
    parity("pool2x2_sum_f32", dx, _nhwc(dref), rel=1e-6, maxabs=1e-6)
    s = torch.empty_like(xs)
    ops.add_f16(xs, dx, s)
    assert torch.equal(s, xs + dx)
    p_, t_ = torch.randn(B, 4, H, W, generator=g).to(dev), torch.randn(B, 4, H, W, generator=g).to(dev)
    dp, loss, scale = torch.empty_like(p_), torch.zeros(1, device=dev), torch.full((1,), 8.0, device=dev)
    ops.mse_loss(p_, t_, dp, loss, scale)
    pr = p_.double().cpu().requires_grad_(True)
    lr = F.mse_loss(pr, t_.double().cpu())
    lr.backward()
    assert abs(loss.item() - lr.item()) < 1e-6 * abs(lr.item()) + 1e-9
    parity("mse_loss_f32 grad", dp, pr.grad * 8.0, rel=1e-6, maxabs=1e-6)


# ------------------------------------------------------------------------------------------------ executors in the fp32 mode
def _small_unet_f32(B, hw, D, sd2=False, seed=0):
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd.unet import HipUNet, UNetGeometry
    torch.manual_seed(seed)
    cfg = UNetConfig.tiny(D)
    if sd2:
        cfg.use_linear_projection = True
        cfg.num_heads = (1, 2, 2, 2)
    ref = UNet2DCondition(cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
    geo = UNetGeometry(block_out_channels=cfg.block_out_channels, num_heads=cfg.num_heads, cross_attention_dim=D,
                       cross_attn_levels=cfg.cross_attn_levels, use_linear_projection=cfg.use_linear_projection)
    hip = HipUNet(geo, ref.state_dict(), B, hw, hw, text_len=77, device=dev, dtype=torch.float32)
    return ref, hip


@pytest.mark.parametrize("sd2", [False, True])
def test_unet_f32_forward_and_dgrad_backward_vs_oracle(sd2):
    """the whole UNet executor with dtype = float32 (no-AMP mode, train_textboost.py:930-939) against the fp32 oracle: 1e-4 is the bar the
    mode is held to (rel-L2, max-abs and worst channel); measured ~1e-6."""
    B, hw, D = 2, 16, 64
    ref, hip = _small_unet_f32(B, hw, D, sd2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.tensor([17, 801])
    ehs = torch.randn(B, 77, D, generator=g).requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, hw, hw, generator=g)
    pred_ref.backward(dpred)
    pred = hip.forward(x.to(dev), t.to(dev), ehs.detach().view(B * 77, D).to(dev).contiguous())
    assert pred.dtype == torch.float32
    parity("fp32 UNet pred", pred, pred_ref, rel=1e-4, maxabs=1e-4, ch_dim=1, ch_rel=1e-4)
    d_ehs = hip.backward(dpred.to(dev))
    parity("fp32 UNet d_ehs", d_ehs.view(B, 77, D), ehs.grad, rel=1e-4, maxabs=1e-4, ch_dim=2, ch_rel=2e-4)


def test_sd15_unet_full_size_f32_vs_oracle():
    """the README command's arithmetic at full size: SD1.5 UNet, B=1, 64x64 latents, fp32 weights / activations / gradients."""
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from textboost_amd import models
    from textboost_amd.unet import HipUNet
    sd = models.random_state_dict(models.unet_shapes(models.SD15_UNET), 91, device="cpu")
    with torch.device("meta"):
        ref = UNet2DCondition(UNetConfig.sd15())
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    for p in ref.parameters():
        p.requires_grad_(False)
    B = 1
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 4, 64, 64, generator=g)
    t = torch.tensor([611])
    ehs = torch.randn(B, 77, 768, generator=g).requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, 64, 64, generator=g)
    pred_ref.backward(dpred)
    hip = HipUNet(models.SD15_UNET, {k: v.to(dev) for k, v in sd.items()}, B, 64, 64, device=dev, dtype=torch.float32)
    pred = hip.forward(x.to(dev), t.to(dev), ehs.detach().view(B * 77, 768).to(dev).contiguous())
    parity("fp32 SD1.5 UNet pred", pred, pred_ref, rel=1e-4, maxabs=1e-4, ch_dim=1, ch_rel=1e-4)
    d_ehs = hip.backward(dpred.to(dev))
    parity("fp32 SD1.5 UNet d_ehs", d_ehs.view(B, 77, 768), ehs.grad, rel=1e-4, maxabs=2e-4, ch_dim=2, ch_rel=5e-4)


def _encoders_f32(B, D, r=4, n_added=3, seed=0, act="quick_gelu"):
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder, add_tokens
    from oracle import train_step as ts
    from textboost_amd.text_encoder import CLIPGeometry, HipTextEncoder
    torch.manual_seed(seed)
    ccfg = CLIPTextCfg.tiny(D)
    ccfg.act = act
    base = TextBoostEncoder(ccfg, r=0)
    with torch.no_grad():
        for n, p in base.named_parameters():
            if "ln" in n or n.endswith("bias"):
                p.add_(torch.randn_like(p) * 0.1)
        base.token_embedding.weight.mul_(0.5)
        null = base.transformer(torch.tensor([[49406] + [49407] * 76]))[0]
    base.set_null_embedding(null)
    teacher = ts.make_teacher(base)
    student = TextBoostEncoder(ccfg, r=r)
    student.load_state_dict(base.state_dict(), strict=False)
    student.set_null_embedding(null)
    with torch.no_grad():
        for n, p in student.named_parameters():
            if "lora_B" in n:
                p.normal_(std=0.05)
    added = add_tokens(student, [100, 200, 300][:n_added])
    geo = CLIPGeometry(hidden_size=D, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers, num_heads=ccfg.num_heads, act=act)
    sd = {hf: dict(base.named_parameters())[ours].detach() for ours, hf in base.hf_key_map().items()}
    hip = HipTextEncoder(geo, sd, B, mode="fp32", lora_rank=r, n_slots=2, device=dev, seed=0)
    hip.set_null_embedding(null)
    hip.add_tokens([100, 200, 300][:n_added])
    for i, layer in enumerate(student.layers if r else []):
        hip.lora_A[i].copy_(torch.cat([layer.q.lora_A, layer.k.lora_A, layer.v.lora_A]).detach())
        hip.lora_B[i].copy_(torch.cat([layer.q.lora_B, layer.k.lora_B, layer.v.lora_B]).detach())
    hip_teacher = HipTextEncoder(geo, sd, B, mode="fp32", lora_rank=0, device=dev)
    hip_teacher.set_null_embedding(null)
    return student, teacher, hip, hip_teacher, added, null


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_encoder_f32_forward_backward_vs_oracle(act):
    """the trainable encoder in the fp32 mode (fp32 LoRA products as exact-fp32 GEMMs) and the fp32 KPL teacher rows in the same pass"""
    from oracle import train_step as ts
    B, D = 3, 64
    student, teacher, hip, hip_teacher, added, null = _encoders_f32(B, D, act=act)
    g = torch.Generator().manual_seed(2)
    ids = ts.synthetic_ids(B, added, g)
    ids[1, 1:] = 49407
    pids = ts.synthetic_ids(B, added, g, prior=True)
    out_ref = student(ids)
    R = torch.randn(B, 77, D, generator=g)
    (out_ref * R).sum().backward()
    with torch.no_grad():
        t_ref = teacher(pids)
    hip.pack_lora()
    merged = hip.forward(ids.to(dev), slot=0, extra_ids=pids.to(dev), extra_table=hip_teacher.token_table)
    out, t_out = merged[:B * 77], merged[B * 77:]
    parity("fp32 encoder hidden states", out.view(B, 77, D), out_ref, rel=1e-5, maxabs=2e-5, ch_dim=2, ch_rel=1e-4)
    parity("fp32 teacher rows", t_out.view(B, 77, D), t_ref, rel=1e-5, maxabs=2e-5)
    assert torch.equal(out.view(B, 77, D)[1].cpu(), null)
    hip.zero_grad()
    hip.backward(R.view(B * 77, D).to(dev).contiguous(), slot=0)
    gA = torch.stack([torch.cat([l.q.lora_A.grad, l.k.lora_A.grad, l.v.lora_A.grad]) for l in student.layers])
    gB = torch.stack([torch.cat([l.q.lora_B.grad, l.k.lora_B.grad, l.v.lora_B.grad]) for l in student.layers])
    parity("fp32 encoder grad lora_A", hip.grad_A, gA, rel=1e-4, maxabs=1e-4, ch_dim=0, ch_rel=1e-4)
    parity("fp32 encoder grad lora_B", hip.grad_B, gB, rel=1e-4, maxabs=1e-4, ch_dim=0, ch_rel=1e-4)
    parity("fp32 encoder grad added rows", hip.grad_added, student.token_embedding.weight.grad[added], rel=1e-4, maxabs=1e-4)


@pytest.mark.parametrize("kpl_type,prediction_type", [("cos", "epsilon"), ("mse", "v_prediction")])
def test_full_step_f32_matches_oracle_elementwise(kpl_type, prediction_type):
    """Two optimizer steps of the no-AMP mode (no GradScaler: loss scale 1, nothing is ever skipped) against the fp32 oracle step:
    losses, gradients (1e-4) and the updated parameters."""
    from oracle import train_step as ts
    from textboost_amd.trainer import StepHyper, TextBoostStep
    B, hw, D = 2, 16, 64
    ref_unet, hip_unet = _small_unet_f32(B, hw, D, seed=3)
    student, teacher, hip_te, hip_teacher, added, null = _encoders_f32(B, D, seed=4)
    st_ref = ts.TrainState(student, teacher, ref_unet, added, ts.StepConfig(kpl_type=kpl_type, prediction_type=prediction_type))
    hp = StepHyper(use_grad_scaler=False, init_scale=1.0, kpl_type=kpl_type, prediction_type=prediction_type)
    step = TextBoostStep(hip_unet, hip_te, hip_teacher, hp, (B, 4, hw, hw), device=dev)
    step.external_noise = True
    g = torch.Generator().manual_seed(5)
    te_ref = st_ref.te
    for it in range(2):
        ids, pids = ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True)
        x0, noise = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        A_before = step.te.lora_A.clone()
        out = st_ref.step(x0, noise, t, ids, pids)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        sc = step.scalars()
        assert sc["found_inf"] == 0.0 and sc["loss_scale"] == 1.0
        assert abs(sc["loss_mse"] - out["mse"]) < 1e-5 * abs(out["mse"]) + 1e-7, (sc, out["mse"])
        assert abs(sc["loss_kpl"] - out["kpl"]) < 1e-4 * abs(out["kpl"]) + 1e-7, (sc, out["kpl"])
        L_ = len(te_ref.layers)
        gA = torch.stack([torch.cat(out["g_lora"][6 * l + 0: 6 * l + 6: 2]) for l in range(L_)])
        gB = torch.stack([torch.cat(out["g_lora"][6 * l + 1: 6 * l + 6: 2]) for l in range(L_)])
        clip = min(1.0, 1.0 / (out["lora_grad_norm"] + 1e-6))   # the oracle's g_lora are post-clip
        assert abs(sc["grad_norm"] - out["lora_grad_norm"]) < 1e-4 * out["lora_grad_norm"]
        parity(f"fp32 step {it} grad lora_A", step.te.grad_A * clip, gA, rel=1e-4, maxabs=1e-4)
        parity(f"fp32 step {it} grad lora_B", step.te.grad_B * clip, gB, rel=1e-4, maxabs=1e-4)
        parity(f"fp32 step {it} grad added rows", step.te.grad_added, out["g_emb_added"], rel=1e-4, maxabs=1e-4)
        w, wr = step.te.token_table.cpu(), te_ref.token_embedding.weight.detach()
        torch.testing.assert_close(w[:49408], wr[:49408], rtol=1e-6, atol=1e-7)
        parity(f"fp32 step {it} updated added rows", w[added], wr[added], rel=2e-4, maxabs=5e-4)
        A_ref = torch.stack([torch.cat([l.q.lora_A, l.k.lora_A, l.v.lora_A]) for l in te_ref.layers]).detach()
        assert (step.te.lora_A.cpu() - A_ref).abs().max().item() < 2e-5           # < lr / 2: no element moved the other way
        assert not torch.equal(step.te.lora_A, A_before)
    assert step.scalars()["opt_steps"] == 2.0


def test_cli_readme_command_runs_in_fp32(tmp_path):
    """The reference's README command (README.md:58-76) passes no --mixed_precision: everything runs in fp32 (train_textboost.py:298-308,
    :930-939), no GradScaler.  4 steps on synthetic latents through the CLI: the step is captured in a HIP graph, the loss scale stays 1,
    no step is skipped, the reference output layout is written, and a second run with the same seed reproduces the weights bit for bit."""
    import os
    import sys
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T

    def run(out):
        args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", out, "--train_batch_size", "2",
                             "--resolution", "128", "--max_train_steps", "4", "--checkpointing_steps", "4", "--placeholder_token", "<dog>",
                             "--augment_inversion", "--lora_rank", "4", "--learning_rate", "5e-5", "--emb_learning_rate", "1e-3",
                             "--seed", "42", "--kpl_weight", "0.1"])
        assert args.mixed_precision is None
        T.main(args)
        return load_file(os.path.join(out, "text_encoder", "adapter_model.safetensors")), torch.load(os.path.join(out, "dog.bin"))

    sd1, tok1 = run(str(tmp_path / "a"))
    log = open(os.path.join(str(tmp_path / "a"), "training.log")).read()
    assert "precision fp32" in log and " scale 1 " in log   # the no-AMP mode ran, loss scale stayed 1
    assert len(sd1) == 72 and any(v.abs().max() > 0 for k, v in sd1.items() if "lora_B" in k)
    assert os.path.isdir(os.path.join(str(tmp_path / "a"), "checkpoint-4"))
    sd2, tok2 = run(str(tmp_path / "b"))
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    assert torch.equal(tok1["<dog>"], tok2["<dog>"])


def test_unet_crossattn_kv_lora_step_matches_oracle(tmp_path):
    """--unet_params_to_train crossattn_kv (train_textboost.py:712-721, :838-841): rank-r adapters on all 16 attn2.to_k / to_v,
    gradients off the hoisted K/V operand, third (unclipped) AdamW group.  Two fp32 optimizer steps against the oracle: adapter gradients,
    the extra d(ehs) term through the text-encoder gradients, the updated adapters; plus the checkpoint round trip and <out>/unet/."""
    from oracle import train_step as ts
    from textboost_amd import checkpoint as ckpt
    from textboost_amd.trainer import StepHyper, TextBoostStep
    B, hw, D, r = 2, 16, 64, 4
    ref_unet, hip_unet = _small_unet_f32(B, hw, D, seed=3)
    adapters = ref_unet.add_crossattn_kv_adapters(r)
    hip_unet.enable_kv_lora(r, seed=0)
    with torch.no_grad():   # non-zero B so that every gradient path is exercised; copy the oracle's adapters into the flat layout
        for ps in adapters.values():
            ps[1].normal_(std=0.05)
            ps[3].normal_(std=0.05)
    for l, (p, C) in enumerate(hip_unet.xattn):
        kA, kB, vA, vB = adapters[p]
        ko = hip_unet.kv_off[p]
        hip_unet.kv_lora_A[l, :r].copy_(kA.detach()); hip_unet.kv_lora_A[l, r:].copy_(vA.detach())
        hip_unet.kv_lora_B[ko:ko + C].copy_(kB.detach()); hip_unet.kv_lora_B[ko + C:ko + 2 * C].copy_(vB.detach())
    student, teacher, hip_te, hip_teacher, added, null = _encoders_f32(B, D, seed=4)
    unet_params = [q for ps in adapters.values() for q in ps]
    st_ref = ts.TrainState(student, teacher, ref_unet, added, ts.StepConfig(), unet_lora=unet_params)
    hp = StepHyper(use_grad_scaler=False, init_scale=1.0)
    step = TextBoostStep(hip_unet, hip_te, hip_teacher, hp, (B, 4, hw, hw), device=dev)
    assert step.n_unet == sum(q.numel() for q in unet_params)
    step.external_noise = True
    g = torch.Generator().manual_seed(7)

    def flat(ps_list, which):  # oracle tensors -> the executor's layout
        A = torch.stack([torch.cat([ps_list[4 * l + 0], ps_list[4 * l + 2]]) for l in range(len(hip_unet.xattn))])
        Bm = torch.cat([torch.cat([ps_list[4 * l + 1], ps_list[4 * l + 3]]) for l in range(len(hip_unet.xattn))])
        return A if which == "A" else Bm

    for it in range(2):
        ids, pids = ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True)
        x0, noise = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        out = st_ref.step(x0, noise, t, ids, pids)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t)
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        sc = step.scalars()
        assert abs(sc["loss_mse"] - out["mse"]) < 1e-5 * abs(out["mse"]) + 1e-7
        parity(f"step {it} grad UNet lora_A", hip_unet.kv_grad_A, flat(out["g_unet"], "A"), rel=1e-4, maxabs=1e-4)
        parity(f"step {it} grad UNet lora_B", hip_unet.kv_grad_B, flat(out["g_unet"], "B"), rel=1e-4, maxabs=1e-4)
        parity(f"step {it} grad added rows (through the adapters' d_ehs term)", step.te.grad_added, out["g_emb_added"], rel=1e-4, maxabs=1e-4)
        cur = [q.detach() for q in unet_params]
        assert (hip_unet.kv_lora_A.cpu() - flat(cur, "A")).abs().max().item() < 2e-5      # < lr / 2
        assert (hip_unet.kv_lora_B.cpu() - flat(cur, "B")).abs().max().item() < 2e-5
    # checkpoint round trip + the final unet/ directory
    ckpt.save_trainer_state(step, str(tmp_path / "checkpoint-2"))
    A0, B0, m0 = hip_unet.kv_lora_A.clone(), hip_unet.kv_lora_B.clone(), step.m_unet.clone()
    hip_unet.kv_lora_A.zero_(); hip_unet.kv_lora_B.zero_(); step.m_unet.zero_()
    ckpt.load_trainer_state(step, str(tmp_path / "checkpoint-2"))
    assert torch.equal(hip_unet.kv_lora_A, A0) and torch.equal(hip_unet.kv_lora_B, B0) and torch.equal(step.m_unet, m0)
    ckpt.save_unet_adapters(hip_unet, str(tmp_path / "unet"), "base")
    from safetensors.torch import load_file
    sd = load_file(str(tmp_path / "unet" / "adapter_model.safetensors"))
    # peft's adapter layout: adapter name stripped, `base_model.model.` prefix -- what set_peft_model_state_dict accepts; round trip
    assert len(sd) == 64 and sd["base_model.model.mid_block.attentions.0.transformer_blocks.0.attn2.to_v.lora_A.weight"].shape == (r, D)
    assert not any(".default." in k for k in sd)
    hip_unet.kv_lora_A.zero_(); hip_unet.kv_lora_B.zero_()
    ckpt.load_unet_peft_adapter_state_dict(hip_unet, {k: v.cuda() for k, v in sd.items()})
    assert torch.equal(hip_unet.kv_lora_A, A0) and torch.equal(hip_unet.kv_lora_B, B0)
    #  the in-model layout of earlier rounds' files still loads; a mangled / foreign layout is refused by name, not by a later KeyError
    old_layout = {k[len("base_model.model."):].replace(".weight", ".default.weight"): v.cuda() for k, v in sd.items()}
    hip_unet.kv_lora_A.zero_(); hip_unet.kv_lora_B.zero_()
    ckpt.load_unet_peft_adapter_state_dict(hip_unet, old_layout)
    assert torch.equal(hip_unet.kv_lora_A, A0) and torch.equal(hip_unet.kv_lora_B, B0)
    for bad in ({"unet." + k: v for k, v in sd.items()}, {k.replace(".weight", ".default.weight"): v for k, v in sd.items()}):
        with pytest.raises(KeyError, match="adapter"):
            ckpt.load_unet_peft_adapter_state_dict(hip_unet, bad)
    import json
    cfg = json.load(open(tmp_path / "unet" / "adapter_config.json"))
    assert set(cfg) == set(ckpt.adapter_config(r, "base"))   # a pure LoraConfig: no extra keys


def test_validation_sampler_unet_gets_the_crossattn_kv_adapters_folded_in():
    """log_validation samples with the TRAINED unet (train_textboost.py:453-531); the validation sampler is a separate fp16
    UNet without adapter support, so under --unet_params_to_train crossattn_kv it loads attn2.to_k / to_v with the adapters folded in
    (HipUNet.merged_kv_weight -> load_kv_weight).  The fp32 training UNet WITH adapters, the fp32 UNet with merged weights and NO adapters
    (1e-5: the same product re-associated), and the fp16 sampler UNet with merged weights (fp16 tolerance) predict the same noise; without
    the fold the fp16 UNet is measurably off."""
    from textboost_amd.unet import HipUNet
    B, hw, D, r = 2, 16, 64, 4
    ref, hip = _small_unet_f32(B, hw, D, seed=5)
    hip.enable_kv_lora(r, seed=1)
    hip.kv_lora_B.normal_(std=0.3)           # (trained adapters: B is zero at initialisation)
    hip.pack_kv_lora()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 4, hw, hw, generator=g).to(dev)
    t = torch.tensor([17, 801]).to(dev)
    ehs = torch.randn(B * 77, D, generator=g).to(dev)
    pred_train = hip.forward(x, t, ehs).clone()
    merged = hip.merged_kv_weight()
    assert merged.dtype == torch.float32 and (merged - hip.P["kv_all.w"]).abs().max() > 1e-3
    plain32 = HipUNet(hip.geo, ref.state_dict(), B, hw, hw, text_len=77, device=dev, dtype=torch.float32)
    pred_base = plain32.forward(x, t, ehs).clone()
    plain32.load_kv_weight(merged)
    parity("fp32 UNet, adapters folded into to_k / to_v", plain32.forward(x, t, ehs), pred_train, rel=1e-5, maxabs=1e-5)
    sampler_unet = HipUNet(hip.geo, ref.state_dict(), B, hw, hw, text_len=77, device=dev)       # the sampler's: fp16, no adapters
    half = sampler_unet.P["kv_all.w"].dtype
    off = rel_err(sampler_unet.forward(x.to(half), t, ehs.to(half)), pred_train)
    sampler_unet.load_kv_weight(merged)
    on = rel_err(sampler_unet.forward(x.to(half), t, ehs.to(half)), pred_train)
    base_gap = rel_err(pred_base, pred_train)
    assert on < 5e-3 and base_gap > 5 * on and off > 5 * on, (on, off, base_gap)


def test_cli_unet_crossattn_kv(tmp_path):
    """the CLI with --unet_params_to_train crossattn_kv (fp32 mode): trains, checkpoints (model_1.safetensors), writes <out>/unet/; the same flag
    under --mixed_precision fp16 has no runnable reference behaviour (:937 casts the adapters to fp16, GradScaler rejects them) and must raise."""
    import os
    import sys
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    out = str(tmp_path / "o")
    base = ["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", out, "--train_batch_size", "1", "--resolution", "128",
            "--max_train_steps", "3", "--checkpointing_steps", "2", "--placeholder_token", "<dog>", "--lora_rank", "4", "--seed", "1",
            "--unet_params_to_train", "crossattn_kv"]
    T.main(T.parse_args(base))
    sd = load_file(os.path.join(out, "unet", "adapter_model.safetensors"))
    assert len(sd) == 64
    assert any(v.abs().max() > 0 for k, v in sd.items() if "lora_B" in k), "the UNet adapters did not train"
    assert os.path.exists(os.path.join(out, "checkpoint-2", "model_1.safetensors"))
    with pytest.raises(NotImplementedError):
        T.main(T.parse_args(base + ["--mixed_precision", "fp16"]))
    with pytest.raises(NotImplementedError):
        T.main(T.parse_args(base[:-1] + ["attn"]))


def test_gradient_accumulation_equals_one_step_on_the_concatenated_batch():
    """--gradient_accumulation_steps 2 (train_textboost.py:1039: accelerate divides each micro loss by G, the optimizer / schedule / step
    counter move on the G-th batch only): two micro batches of B must give the update of ONE step on the 2B concatenation (both losses
    are batch means).  fp32 mode, eager and as the two captured graphs."""
    from oracle import train_step as ts
    from textboost_amd.trainer import StepHyper, TextBoostStep
    hw, D = 16, 64

    def make(B, accum):
        _, hip_unet = _small_unet_f32(B, hw, D, seed=3)
        _, _, hip_te, hip_teacher, added, _ = _encoders_f32(B, D, seed=4)
        st = TextBoostStep(hip_unet, hip_te, hip_teacher, StepHyper(use_grad_scaler=False, init_scale=1.0, grad_accum=accum), (B, 4, hw, hw),
                           device=dev)
        st.external_noise = True
        return st, added

    g = torch.Generator().manual_seed(8)
    big, added = make(4, 1)
    ids, pids = ts.synthetic_ids(4, added, g), ts.synthetic_ids(4, added, g, prior=True)
    x0, noise, t = torch.randn(4, 4, hw, hw, generator=g), torch.randn(4, 4, hw, hw, generator=g), torch.randint(0, 1000, (4,), generator=g)
    big.x0.copy_(x0); big.noise.copy_(noise); big.timesteps.copy_(t); big.input_ids.copy_(ids); big.prior_ids.copy_(pids)
    assert big.step_eager() is True
    for graphs in (False, True):
        acc, _ = make(2, 2)
        if graphs:
            acc.capture(warmup=0)
            assert acc.graph_mode == "micro+tail"
        synced = []
        for h in range(2):
            sl = slice(2 * h, 2 * h + 2)
            acc.x0.copy_(x0[sl]); acc.noise.copy_(noise[sl]); acc.timesteps.copy_(t[sl]); acc.input_ids.copy_(ids[sl]); acc.prior_ids.copy_(pids[sl])
            synced.append(acc.replay() if graphs else acc.step_eager())
        assert synced == [False, True]
        assert acc.scalars()["opt_steps"] == 1.0
        parity("accumulated gradient / G", acc.flat_grad / 2, big.flat_grad, rel=1e-5, maxabs=1e-5)
        assert (acc.te.lora_A - big.te.lora_A).abs().max().item() < 2e-5 and (acc.te.lora_B - big.te.lora_B).abs().max().item() < 2e-5
        parity("updated added rows", acc.te.token_table[acc.te.first_added:], big.te.token_table[big.te.first_added:], rel=1e-5, maxabs=1e-4)


def test_lora_rank_zero_trains_the_added_rows_only():
    """--lora_rank 0 (:700: no adapter is injected; optimizer group 1 is empty): one fp32 step against the oracle without adapters."""
    from oracle import train_step as ts
    from textboost_amd.trainer import StepHyper, TextBoostStep
    B, hw, D = 2, 16, 64
    ref_unet, hip_unet = _small_unet_f32(B, hw, D, seed=3)
    student, teacher, hip_te, hip_teacher, added, null = _encoders_f32(B, D, r=0, seed=4)
    st_ref = ts.TrainState(student, teacher, ref_unet, added, ts.StepConfig())
    step = TextBoostStep(hip_unet, hip_te, hip_teacher, StepHyper(use_grad_scaler=False, init_scale=1.0), (B, 4, hw, hw), device=dev)
    step.external_noise = True
    assert step.n_lora == 0
    g = torch.Generator().manual_seed(9)
    ids, pids = ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True)
    x0, noise, t = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g), torch.randint(0, 1000, (B,), generator=g)
    out = st_ref.step(x0, noise, t, ids, pids)
    step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t); step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
    step.step_eager()
    sc = step.scalars()
    assert sc["opt_steps"] == 1.0 and abs(sc["loss_mse"] - out["mse"]) < 1e-5 * abs(out["mse"]) + 1e-7
    assert abs(sc["loss_kpl"] - out["kpl"]) < 1e-4 * abs(out["kpl"]) + 1e-6   # student == teacher up to the added rows
    parity("r=0 grad added rows", step.te.grad_added, out["g_emb_added"], rel=1e-4, maxabs=1e-4)
    w, wr = step.te.token_table.cpu(), student.token_embedding.weight.detach()
    torch.testing.assert_close(w[:49408], wr[:49408], rtol=1e-6, atol=1e-7)
    parity("r=0 updated added rows", w[added], wr[added], rel=2e-4, maxabs=5e-4)
