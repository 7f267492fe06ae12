"""GPU parity: GroupNorm(+SiLU), LayerNorm, flash attention -- forward and input gradients vs torch fp32."""
import pytest
import torch
import torch.nn.functional as F

from parity import parity

pytestmark = pytest.mark.gpu


def _ops():
    from textboost_amd import ops
    return ops


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize("B,HW,C,silu,extra", [(2, 256, 320, True, 0), (3, 100, 640, False, 64), (1, 64, 1280, True, 0),
                                               (2, 1024, 960, True, 0), (2, 64, 2560, True, 0), (2, 256, 1920, False, 0),
                                               (2, 144, 64, True, 0),
                                               # >= 128 (image, group) slices with 8-aligned groups: the one-pass per-slice kernels
                                               (8, 256, 1280, True, 0), (4, 64, 2560, True, 0), (8, 256, 2560, False, 64), (8, 64, 1280, True, 0),
                                               # large maps, any even group width: the one-pass 1024-thread slice kernels (round 4)
                                               (8, 4096, 320, True, 0), (8, 1024, 640, True, 64), (8, 4096, 640, False, 0), (8, 1024, 1920, True, 0),
                                               (8, 1024, 960, True, 64), (8, 4096, 960, True, 0), (4, 1000, 1280, True, 0), (16, 1024, 320, False, 0)])
def test_groupnorm_fwd_bwd(B, HW, C, silu, extra):
    ops = _ops()
    torch.manual_seed(0)
    M = B * HW
    xbuf = (torch.randn(M, C + extra, device="cuda") * 2 + 0.5).half()
    x = xbuf[:, extra // 2: extra // 2 + C]
    gamma = torch.randn(C, device="cuda") * 0.5 + 1
    beta = torch.randn(C, device="cuda") * 0.3
    y = torch.empty(M, C, device="cuda", dtype=torch.float16)
    stats = torch.empty(B, 32, 2, device="cuda")
    ws = torch.empty(ops.groupnorm_ws(B, HW, C), device="cuda")
    ops.groupnorm_fwd(x, y, gamma, beta, stats, ws, B, HW, C, eps=1e-5, silu=silu)
    xr = x.float().view(B, HW, C).permute(0, 2, 1).contiguous().requires_grad_(True)  # [B, C, HW]
    ref = F.group_norm(xr, 32, gamma, beta, eps=1e-5)
    if silu:
        ref = F.silu(ref)
    parity(f"groupnorm fwd {B}x{HW}x{C}", y.view(B, HW, C), ref.permute(0, 2, 1), rel=2e-3, maxabs=3e-3, ch_dim=2, ch_rel=3e-3)
    dy = torch.randn(M, C, device="cuda").half()
    add = torch.randn(M, C, device="cuda").half()
    dx = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.groupnorm_bwd(dy, x, gamma, beta, stats, dx, ws, B, HW, C, silu=silu, add=add)
    ref.backward(dy.float().view(B, HW, C).permute(0, 2, 1))
    gref = xr.grad.permute(0, 2, 1).reshape(M, C) + add.float()
    parity(f"groupnorm bwd {B}x{HW}x{C}", dx, gref, rel=3e-3, maxabs=4e-3, ch_dim=1, ch_rel=4e-3)
    if B * 32 >= 128 and (C // 32) % 2 == 0:  # a one-pass kernel ran: the two-pass kernels must agree (same fp32 arithmetic, other summation order)
        from textboost_amd import _lib as L
        prev = L.lib().tb_groupnorm_set_variant(0)
        y2, dx2, stats2 = torch.empty_like(y), torch.empty_like(dx), torch.empty_like(stats)
        ops.groupnorm_fwd(x, y2, gamma, beta, stats2, ws, B, HW, C, eps=1e-5, silu=silu)
        ops.groupnorm_bwd(dy, x, gamma, beta, stats2, dx2, ws, B, HW, C, silu=silu, add=add)
        L.lib().tb_groupnorm_set_variant(prev)
        torch.testing.assert_close(stats, stats2, rtol=1e-5, atol=1e-6)
        assert rel_err(y, y2) < 1e-4 and rel_err(dx, dx2) < 1e-4


@pytest.mark.parametrize("M,C,xdt", [(616, 768, torch.float32), (1000, 320, torch.float16), (77, 1280, torch.float16),
                                     (130, 640, torch.float16), (50, 1024, torch.float32), (33, 64, torch.float32)])
def test_layernorm_fwd_bwd(M, C, xdt):
    ops = _ops()
    torch.manual_seed(1)
    x = (torch.randn(M, C, device="cuda") * 1.5 + 0.2).to(xdt)
    gamma = torch.randn(C, device="cuda") * 0.5 + 1
    beta = torch.randn(C, device="cuda") * 0.3
    y = torch.empty(M, C, device="cuda", dtype=torch.float16)
    stats = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, y, gamma, beta, stats, eps=1e-5)
    xr = x.float().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gamma, beta, 1e-5)
    parity(f"layernorm fwd {M}x{C}", y, ref, rel=1e-3, maxabs=2e-3, ch_dim=1, ch_rel=2e-3)
    dy = torch.randn(M, C, device="cuda").half()
    add = torch.randn(M, C, device="cuda").to(xdt)
    dx = torch.empty(M, C, device="cuda", dtype=xdt)
    ops.layernorm_bwd(dy, x, gamma, stats, dx, add=add)
    ref.backward(dy.float())
    tol = 2e-3 if xdt == torch.float16 else 1e-4
    parity(f"layernorm bwd {M}x{C}", dx, xr.grad + add.float(), rel=tol, maxabs=2 * tol, ch_dim=1, ch_rel=2 * tol)
    # optional outputs: fp16 copy of dx, and the fused LoRA down projection of the normalised rows (== tb_lora_down on y)
    dx2 = torch.empty_like(dx); dx16 = torch.zeros(M, C + 8, device="cuda", dtype=torch.float16)[:, :C]
    ops.layernorm_bwd(dy, x, gamma, stats, dx2, add=add, dx16=dx16)
    assert torch.equal(dx2, dx) and torch.equal(dx16, dx.half())
    R = 12
    A = torch.randn(R, C, device="cuda") / 4
    y2 = torch.empty_like(y); t = torch.zeros(M, 64, device="cuda", dtype=torch.float16); t_ref = torch.zeros_like(t)
    ops.layernorm_fwd(x, y2, gamma, beta, stats, eps=1e-5, lora_A=A, t=t)
    ops.lora_down(y, A, t_ref)
    assert torch.equal(y2, y) and t[:, R:].abs().max() == 0
    assert rel_err(t[:, :R], y.float() @ A.half().float().T) < 2e-3 and rel_err(t[:, :R], t_ref[:, :R]) < 2e-3


def ref_attention(q, k, v, H, causal):
    B, Sq, C = q.shape
    hd = C // H
    qh = q.view(B, Sq, H, hd).transpose(1, 2)
    kh = k.view(B, -1, H, hd).transpose(1, 2)
    vh = v.view(B, -1, H, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * hd ** -0.5
    if causal:
        s = s + torch.full_like(s[0, 0], float("-inf")).triu(1)
    p = torch.softmax(s, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, C), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,H,Sq,Skv,hd,causal", [
    (2, 8, 256, 256, 40, False),    # UNet L0-style self attention (hd 40 -> padded tiles)
    (1, 8, 320, 320, 80, False),    # L1 (ragged vs 128 / 64 tiles)
    (2, 4, 64, 64, 160, False),     # L2/L3
    (2, 8, 200, 77, 40, False),     # cross attention, 77 text tokens
    (2, 8, 1024, 77, 40, False),    # cross attention, long query range -> several q slices per key block
    (3, 12, 77, 77, 64, True),      # CLIP causal
    (2, 2, 100, 100, 32, False),    # tiny-config head dims
    (1, 2, 130, 70, 128, False),
    (1, 3, 77, 77, 64, False),
])
def test_attention_fwd_bwd(B, H, Sq, Skv, hd, causal):
    ops = _ops()
    torch.manual_seed(2)
    C = H * hd
    # q/k/v as column slices of one fused buffer when self-attention (like the fused qkv GEMM output)
    if Sq == Skv:
        qkv = torch.randn(B * Sq, 3 * C, device="cuda").half()
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        q = torch.randn(B * Sq, C, device="cuda").half()
        kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
        k, v = kv[:, :C], kv[:, C:]
    o = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, Sq, device="cuda")
    ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd, causal=causal)
    qr, kr, vr = [t.float().reshape(B, -1, C).requires_grad_(True) for t in (q, k, v)]
    oref, lref = ref_attention(qr, kr, vr, H, causal)
    # rel-L2 AND max-abs (relative to the largest reference magnitude) AND the worst head-dim channel (tests/parity.py)
    parity(f"attention O {B}x{H}x{Sq}x{Skv}x{hd}", o.view(B, Sq, C), oref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-3)
    assert (lse - lref).abs().max().item() < 2e-3
    do = torch.randn(B * Sq, C, device="cuda").half()
    delta = torch.empty(B, H, Sq, device="cuda")
    dq = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
    dkv = torch.empty(B * Skv, 2 * C, device="cuda", dtype=torch.float16)
    dk, dv = dkv[:, :C], dkv[:, C:]
    ws = torch.empty(8 * 2 * B * Skv * C, device="cuda") if Skv <= 128 else None   # exercises the split-q path
    ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Sq, Skv, hd, causal=causal, ws=ws)
    oref.backward(do.float().view(B, Sq, C))
    parity("attention dQ", dq.view(B, Sq, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
    parity("attention dK", dk.reshape(B, Skv, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
    parity("attention dV", dv.reshape(B, Skv, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 2, 256, 128), (1, 2, 256, 192), (2, 3, 512, 320), (1, 8, 1024, 1024), (2, 8, 4096, 4096)])
def test_attention_fwd_software_pipelined_kernel(B, H, Sq, Skv):
    """attn_fwd_il_kernel (hd = 40, Sq % 256 == 0, Skv % 64 == 0, >= 2 KV tiles: the SD1.x 64x64-map self-attention shape and its smaller
    relatives): two query groups per wave run half an iteration apart, so the pipeline's prologue (2 tiles), the 4-slot ring wrapping,
    and the epilogue are all exercised; keys are spiked in BOTH half-wave key sets of a late tile (the running-max exchange between the
    half-waves) to force the re-base branch (cdna guide rule 26).  Against fp32 attention: rel-L2 / max-abs / worst channel / LSE."""
    from textboost_amd import _lib as L
    ops = _ops()
    torch.manual_seed(5)
    hd = 40
    C = H * hd
    q = torch.randn(B * Sq, C, device="cuda").half()
    kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
    k, v = kv[:, :C], kv[:, C:]
    kt = Skv - 64 if Skv > 128 else 64          # a late tile
    for h in range(H):
        k.view(B, Skv, C)[0, kt + 3, h * hd:(h + 1) * hd] = q.view(B, Sq, C)[0, 7 + h, h * hd:(h + 1) * hd] * 3    # key 3 of the tile: low half-wave
        k.view(B, Skv, C)[0, kt + 22, h * hd:(h + 1) * hd] = q.view(B, Sq, C)[0, 40 + h, h * hd:(h + 1) * hd] * 3  # key 22: high half-wave
    oref, lref = ref_attention(q.float().reshape(B, Sq, C), k.float().reshape(B, Skv, C), v.float().reshape(B, Skv, C), H, False)
    res = []
    for variant in (1, 1 | 1024):   # the pipelined kernel, then the LDS-DMA kernel it replaces for this shape
        L.lib().tb_attention_set_variant(variant)
        o = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
        lse = torch.empty(B, H, Sq, device="cuda")
        ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
        res.append((o, lse))
    L.lib().tb_attention_set_variant(1)
    for o, lse in res:
        assert torch.isfinite(o).all()
        parity(f"pipelined attention O {B}x{H}x{Sq}x{Skv}", o.view(B, Sq, C), oref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-3)
        assert ((lse - lref).abs() / lref.abs().clamp_min(1.0)).max().item() < 2e-3
    assert rel_err(res[0][0], res[1][0]) < 1e-3


@pytest.mark.parametrize("B,H,Sq,Skv,hd", [(2, 8, 4096, 77, 40), (8, 8, 4096, 77, 40), (2, 8, 1024, 77, 40), (1, 5, 1152, 77, 64), (2, 4, 1024, 50, 40),
                                           (1, 3, 2048, 96, 24), (1, 2, 1280, 65, 8), (2, 8, 1024, 77, 80), (1, 4, 1152, 77, 96)])
def test_attention_fwd_short_key_kernel(B, H, Sq, Skv, hd):
    """attn_xs_fwd_kernel (round 4): cross-attention on the prompt (33 .. 96 keys, hd <= 64, >= 1024 queries) with all keys staged once, K held in
    registers and 1 / 2 / 4 query tiles per wave, against fp32 attention and against the general flash kernel it replaces for these shapes
    (tb_attention_set_variant bit 16384): ragged key counts (50, 65, 77, 96), head dims with and without padding, one dominant key per head."""
    from textboost_amd import _lib as L
    ops = _ops()
    torch.manual_seed(11)
    C = H * hd
    q = torch.randn(B * Sq, C, device="cuda").half()
    kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
    k, v = kv[:, :C], kv[:, C:]
    for h in range(H):   # a dominant key in the last (partly masked) 32-key tile
        k.view(B, Skv, C)[0, Skv - 2, h * hd:(h + 1) * hd] = q.view(B, Sq, C)[0, 9 + h, h * hd:(h + 1) * hd] * 3
    oref, lref = ref_attention(q.float().reshape(B, Sq, C), k.float().reshape(B, Skv, C), v.float().reshape(B, Skv, C), H, False)
    res = []
    prev = L.lib().tb_attention_set_variant(1)
    try:
        for variant in (1, 1 | 16384):
            L.lib().tb_attention_set_variant(variant)
            o = torch.full((B * Sq, C + 8), 5.0, device="cuda", dtype=torch.float16)[:, :C]
            lse = torch.empty(B, H, Sq, device="cuda")
            ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
            assert torch.isfinite(o).all()
            parity(f"short-key attention O {B}x{H}x{Sq}x{Skv}x{hd} variant {variant}", o.reshape(B, Sq, C), oref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-3)
            assert ((lse - lref).abs() / lref.abs().clamp_min(1.0)).max().item() < 2e-3    # (the dominant key's score is ~20: relative)
            res.append((o.clone(), lse))
    finally:
        L.lib().tb_attention_set_variant(prev)
    assert rel_err(res[0][0], res[1][0]) < 1e-3 and (res[0][1] - res[1][1]).abs().max().item() < 5e-4


@pytest.mark.parametrize("B,H,Sq,Skv,hd", [(2, 8, 4096, 77, 40), (2, 8, 1024, 77, 40), (1, 5, 1152, 77, 64), (2, 4, 1024, 50, 40), (1, 3, 2048, 96, 24)])
def test_attention_bwd_short_key_dq_kernel(B, H, Sq, Skv, hd):
    """attn_xs_bwd_dq_kernel (round 4): the dQ half of the cross-attention backward for 33 .. 96 keys (K / V staged once and held as register
    fragments, two query tiles per wave requested up front; also publishes delta for the dK / dV launch) against autograd and against the general
    kernels (tb_attention_set_variant bit 16384): dQ, dK, dV and delta."""
    from textboost_amd import _lib as L
    ops = _ops()
    torch.manual_seed(12)
    C = H * hd
    q = torch.randn(B * Sq, C, device="cuda").half()
    kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
    k, v = kv[:, :C], kv[:, C:]
    do = torch.randn(B * Sq, C, device="cuda").half()
    qr, kr, vr = [t.float().reshape(B, -1, C).requires_grad_(True) for t in (q, k, v)]
    oref, _ = ref_attention(qr, kr, vr, H, False)
    oref.backward(do.float().view(B, Sq, C))
    res = []
    prev = L.lib().tb_attention_set_variant(1)
    try:
        for variant in (1 | 65536, 1 | 65536 | 16384):     # (65536: not round 5's one-launch backward, tested below)
            L.lib().tb_attention_set_variant(variant)
            o = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
            lse = torch.empty(B, H, Sq, device="cuda")
            ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
            delta = torch.full((B, H, Sq), 7.0, device="cuda")
            dq = torch.full((B * Sq, C + 8), 5.0, device="cuda", dtype=torch.float16)[:, :C]
            dkv = torch.empty(B * Skv, 2 * C, device="cuda", dtype=torch.float16)
            ws = torch.empty(8 * 2 * B * Skv * C, device="cuda")
            ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dkv[:, :C], dkv[:, C:], B, H, Sq, Skv, hd, ws=ws)
            parity(f"short-key dQ variant {variant}", dq.reshape(B, Sq, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            parity(f"short-key dK variant {variant}", dkv[:, :C].reshape(B, Skv, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            parity(f"short-key dV variant {variant}", dkv[:, C:].reshape(B, Skv, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            dref = (do.float() * o.float()).view(B, Sq, H, hd).sum(-1).permute(0, 2, 1)
            assert (delta - dref).abs().max().item() < 2e-3 * dref.abs().max().item() + 1e-4
            res.append(dq.clone())
    finally:
        L.lib().tb_attention_set_variant(prev)
    assert rel_err(res[0], res[1]) < 1e-3


@pytest.mark.parametrize("B,H,Sq,Skv,hd", [(2, 8, 4096, 77, 40), (8, 8, 4096, 77, 40), (8, 8, 1024, 77, 80), (2, 8, 1024, 77, 40), (1, 5, 1152, 77, 64),
                                           (2, 4, 1024, 50, 40), (1, 3, 2048, 96, 24), (2, 5, 2304, 77, 64)])
def test_attention_bwd_cross_attention_in_one_launch(B, H, Sq, Skv, hd):
    """attn_xs_bwd_kernel (round 5; diffusers attn2 backward, train_textboost.py:1108): dQ, dK and dV of the cross-attention on the prompt tokens in
    ONE launch over query slices (+ the finalize of the dK / dV slices) -- Q and dO staged once per tile, delta = rowsum(P * dP) so the attention
    output is not read -- against autograd and against the three launches it replaces (tb_attention_set_variant bit 65536).  `delta` is not an
    output of this path (nothing downstream reads it)."""
    from textboost_amd import _lib as L
    ops = _ops()
    torch.manual_seed(14)
    C = H * hd
    q = torch.randn(B * Sq, C, device="cuda").half()
    kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
    k, v = kv[:, :C], kv[:, C:]
    do = torch.randn(B * Sq, C, device="cuda").half()
    qr, kr, vr = [t.float().reshape(B, -1, C).requires_grad_(True) for t in (q, k, v)]
    oref, _ = ref_attention(qr, kr, vr, H, False)
    oref.backward(do.float().view(B, Sq, C))
    res = []
    prev = L.lib().tb_attention_set_variant(1)
    try:
        for variant in (1 | 131072, 1 | 65536):     # (131072: the one-launch kernel also for hd = 80, where the default keeps the three launches)
            L.lib().tb_attention_set_variant(variant)
            o = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
            lse = torch.empty(B, H, Sq, device="cuda")
            ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
            delta = torch.full((B, H, Sq), 7.0, device="cuda")
            dqb = torch.full((B * Sq, C + 8), 5.0, device="cuda", dtype=torch.float16)
            dq = dqb[:, :C]
            dkv = torch.empty(B * Skv, 2 * C, device="cuda", dtype=torch.float16)
            ws = torch.empty(16 * 2 * B * Skv * C, device="cuda")
            if variant == (1 | 131072) and hd in (40, 64, 80):
                o.fill_(float("nan"))     # the one-launch path must not read the attention output
            ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dkv[:, :C], dkv[:, C:], B, H, Sq, Skv, hd, ws=ws)
            assert (dqb[:, C:] == 5.0).all()
            parity(f"cross-attention dQ variant {variant}", dq.reshape(B, Sq, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            parity(f"cross-attention dK variant {variant}", dkv[:, :C].reshape(B, Skv, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            parity(f"cross-attention dV variant {variant}", dkv[:, C:].reshape(B, Skv, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
            res.append((dq.clone(), dkv.clone()))
    finally:
        L.lib().tb_attention_set_variant(prev)
    assert rel_err(res[0][0], res[1][0]) < 1.5e-3 and rel_err(res[0][1], res[1][1]) < 1.5e-3


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a late KV tile (cdna guide rule 26): spike one key against one query."""
    ops = _ops()
    torch.manual_seed(3)
    B, H, S, hd = 1, 2, 256, 64
    C = H * hd
    q = torch.randn(B * S, C, device="cuda").half()
    k = torch.randn(B * S, C, device="cuda").half()
    v = torch.randn(B * S, C, device="cuda").half()
    k[200] = q[5] * 4  # key 200 (4th tile) dominates query 5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    oref, _ = ref_attention(q.float().view(B, S, C), k.float().view(B, S, C), v.float().view(B, S, C), H, False)
    assert (o.float().view(B, S, C) - oref).abs().max().item() < 1e-2


@pytest.mark.parametrize("B,H,Sq,Skv,hd", [(1, 2, 128, 64, 40), (1, 2, 128, 128, 40), (2, 3, 256, 192, 40), (1, 8, 1024, 1024, 40),
                                            (1, 2, 128, 64, 80), (2, 3, 256, 320, 80), (1, 8, 1024, 1024, 80)])
def test_attention_fwd_lds_dma_kernel(B, H, Sq, Skv, hd):
    """the LDS-DMA staged forward kernel (hd = 40 / 80, Sq % 128 == 0, Skv % 64 == 0): 1, 2, 3 and many KV tiles (the 3-slot ring wraps),
    bit-comparable statistics with the register-staged kernel, and the forced re-base branch (cdna guide rule 26)."""
    from textboost_amd import _lib as L
    ops = _ops()
    torch.manual_seed(4)
    C = H * hd
    qkv = torch.randn(B * max(Sq, Skv), 3 * C, device="cuda").half()
    q, k, v = qkv[:B * Sq, :C], qkv[:B * Skv, C:2 * C].clone(), qkv[:B * Skv, 2 * C:]
    if Skv >= 192:
        k.view(B, Skv, C)[0, 150, :hd] = q.view(B, Sq, C)[0, 7, :hd] * 4  # key 150 (3rd tile) dominates query 7 of head 0: re-base at a late tile
    outs = []
    for variant in (1, 0):
        L.lib().tb_attention_set_variant(variant)
        o = torch.empty(B * Sq, C, device="cuda", dtype=torch.float16)
        lse = torch.empty(B, H, Sq, device="cuda")
        ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
        outs.append((o, lse))
    L.lib().tb_attention_set_variant(1)
    oref, lref = ref_attention(q.float().reshape(B, Sq, C), k.float().reshape(B, Skv, C), v.float().reshape(B, Skv, C), H, False)
    for o, lse in outs:
        parity(f"attention O (LDS-DMA kernel) {B}x{H}x{Sq}x{Skv}x{hd}", o.view(B, Sq, C), oref, rel=2e-3, maxabs=4e-3, ch_dim=2, ch_rel=3e-3)
        assert ((lse - lref).abs() / lref.abs().clamp_min(1.0)).max().item() < 2e-3


@pytest.mark.parametrize("B,H,S,spread", [(2, 8, 512, 1.0), (1, 8, 4096, 1.0), (2, 8, 1024, 4.0)])
def test_attention_forward_fp8_pv(B, H, S, spread):
    """BASELINE.json configs[4]: the opt-in e4m3 P.V forward of the hd = 40 self-attention shape (tb_attn_desc.fp8_ws) against fp32
    attention.  Stated tolerance: rel-L2 <= 5e-2 on O; LSE (what the fp16 backward rebuilds the softmax from) within 2e-3, it is computed
    from the exact fp32 row sum -- the fp16 kernel of the same shape is held to 2e-3 / 2e-3 above.  e4m3 keeps 3 mantissa bits: every probability and every scaled V entry carries a rounding error of
    up to 2^-4 (3.6 % rms); on random V the sum over keys is itself a random sum, so the relative error of O does not average down
    (measured 3.6e-2 on unit-normal data).  `spread` scales the logits: peaked rows (few keys carry the weight) are the worst case."""
    ops = _ops()
    torch.manual_seed(3)
    hd, C = 40, H * 40
    qkv = torch.randn(B * S, 3 * C, device="cuda")
    qkv[:, :C] *= spread
    qkv[:, 2 * C:] *= 3.0                       # V away from unit scale: exercises the per-(batch, head) scale
    qkv = qkv.half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ws = ops.attention_fp8_workspace(B, H, S, "cuda")
    o8 = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
    lse8 = torch.empty(B, H, S, device="cuda")
    ops.attention_fwd(q, k, v, o8, lse8, B, H, S, S, hd, fp8_ws=ws)
    o16 = torch.empty_like(o8)
    lse16 = torch.empty_like(lse8)
    ops.attention_fwd(q, k, v, o16, lse16, B, H, S, S, hd)
    oref, lref = ref_attention(*[t.float().reshape(B, S, C) for t in (q, k, v)], H, False)
    e8, e16 = rel_err(o8.view(B, S, C), oref), rel_err(o16.view(B, S, C), oref)
    print(f"[fp8 attention] S={S} spread={spread}: rel-L2 fp8 {e8:.3e} (fp16 kernel {e16:.3e}), max |dLSE| {(lse8 - lref).abs().max().item():.3e}")
    assert e16 < 2e-3
    assert 1e-3 < e8 < 5e-2, "the fp8 path did not run (error equals the fp16 kernel's) or is outside its stated tolerance"
    assert (lse8 - lref).abs().max().item() < 2e-3 * max(1.0, lref.abs().max().item())
    assert torch.isfinite(o8).all()


@pytest.mark.parametrize("B,H,S,hd", [(2, 8, 1024, 40), (8, 8, 512, 40), (1, 8, 4096, 40), (8, 8, 1024, 80), (2, 8, 1024, 80), (1, 5, 9216, 64), (8, 10, 1024, 64)])
def test_attention_backward_dma_staged_dkv(B, H, S, hd):
    """the hd = 40 / 64 / 80 self-attention backward with a ws of 2 * B * H * S floats takes attn_bwd_dkv_dma_kernel (>= 512 key blocks): against fp32
    autograd and against the register-staged kernels (bits 128 / 256 of tb_attention_set_variant switch the DMA dK/dV / dQ kernels off)."""
    ops = _ops()
    from textboost_amd import _lib as L
    torch.manual_seed(4)
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device="cuda").half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    do = torch.randn(B * S, C, device="cuda").half()
    qr, kr, vr = [t.float().reshape(B, S, C).requires_grad_(True) for t in (q, k, v)]
    oref, _ = ref_attention(qr, kr, vr, H, False)
    oref.backward(do.float().view(B, S, C))
    res = []
    old = L.lib().tb_attention_set_variant(1)
    for bits in (1, 1 | 128 | 256):
        L.lib().tb_attention_set_variant(bits)
        delta = torch.empty(B, H, S, device="cuda")
        dqkv = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16)
        ws = torch.empty(2 * B * H * S, device="cuda")
        ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
        res.append(dqkv)
    L.lib().tb_attention_set_variant(old)
    for dqkv in res:
        parity("dQ (DMA / register staged)", dqkv[:, :C].reshape(B, S, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity("dK (DMA / register staged)", dqkv[:, C:2 * C].reshape(B, S, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity("dV (DMA / register staged)", dqkv[:, 2 * C:].reshape(B, S, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
    if (S // 128) * H * B >= 512:
        assert not torch.equal(res[0][:, C:], res[1][:, C:]), "the DMA-staged dK/dV kernel did not run"
    assert not torch.equal(res[0][:, :C], res[1][:, :C]), "the DMA-staged dQ kernel did not run"
    assert rel_err(res[0], res[1]) < 2e-3


@pytest.mark.parametrize("B,H,S", [(8, 8, 1024), (2, 8, 4096), (16, 4, 1024)])
def test_attention_backward_software_pipelined_kernels(B, H, S):
    """attn_bwd_dq_il_kernel / attn_bwd_dkv_il_kernel (hd = 40, >= 512 key blocks: the SD1.x 64x64-map self-attention backward at the metric batch): a three-stage
    pipeline over 32-query halves inside a wave -- prologue, the 4-slot ring wrapping (16 / 64 query tiles), epilogue.  Against fp32 autograd
    (rel-L2 / max-abs / worst channel) and against the LDS-DMA kernel it replaces (bit 2048 of tb_attention_set_variant): same arithmetic in
    the same order, so bit-equal."""
    ops = _ops()
    from textboost_amd import _lib as L
    torch.manual_seed(6)
    hd = 40
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device="cuda").half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    do = torch.randn(B * S, C, device="cuda").half()
    res = []
    old = L.lib().tb_attention_set_variant(1)
    for bits in (1 | 4096, 1 | 2048):   # both pipelined kernels (dQ is opt-in: bit 4096), then the LDS-DMA kernels
        L.lib().tb_attention_set_variant(bits)
        delta = torch.empty(B, H, S, device="cuda")
        dqkv = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16)
        ws = torch.empty(2 * B * H * S, device="cuda")
        ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
        res.append((dqkv, delta))
    L.lib().tb_attention_set_variant(old)
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    res = [r[0] for r in res]
    if S <= 1024:
        qr, kr, vr = [t.float().reshape(B, S, C).requires_grad_(True) for t in (q, k, v)]
        oref, _ = ref_attention(qr, kr, vr, H, False)
        oref.backward(do.float().view(B, S, C))
        parity("pipelined dQ", res[0][:, :C].reshape(B, S, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity("pipelined dK", res[0][:, C:2 * C].reshape(B, S, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity("pipelined dV", res[0][:, 2 * C:].reshape(B, S, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)


@pytest.mark.parametrize("B,H,S,causal", [(3, 12, 77, True), (16, 12, 77, True), (2, 16, 77, True), (1, 3, 77, False), (2, 4, 96, True), (2, 4, 33, True)])
def test_attention_backward_short_sequence_in_one_launch(B, H, S, causal):
    """csrc/attention_small.hip (round 3): dQ, dK, dV of a short hd = 64 sequence -- the CLIP text encoder's causal 77 x 77 self-attention -- from one
    workgroup per (batch, head), against fp32 autograd and against the generic dQ + dK/dV launches (bit 8192 of tb_attention_set_variant)."""
    ops = _ops()
    from textboost_amd import _lib as L
    torch.manual_seed(9)
    hd = 64
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device="cuda").half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd, causal=causal)
    do = torch.randn(B * S, C, device="cuda").half()
    qr, kr, vr = [t.float().reshape(B, S, C).requires_grad_(True) for t in (q, k, v)]
    oref, _ = ref_attention(qr, kr, vr, H, causal)
    oref.backward(do.float().view(B, S, C))
    res = []
    old = L.lib().tb_attention_set_variant(1)
    for bits in (1, 1 | 8192):
        L.lib().tb_attention_set_variant(bits)
        delta = torch.empty(B, H, S, device="cuda")
        dqkv = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16)
        ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, causal=causal)
        res.append(dqkv)
    L.lib().tb_attention_set_variant(old)
    for name, dqkv in (("one launch", res[0]), ("generic", res[1])):
        parity(f"short-sequence dQ ({name})", dqkv[:, :C].reshape(B, S, C), qr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity(f"short-sequence dK ({name})", dqkv[:, C:2 * C].reshape(B, S, C), kr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
        parity(f"short-sequence dV ({name})", dqkv[:, 2 * C:].reshape(B, S, C), vr.grad, rel=4e-3, maxabs=6e-3, ch_dim=2, ch_rel=6e-3)
    assert not torch.equal(res[0], res[1]), "the one-launch kernel did not run"
    assert rel_err(res[0], res[1]) < 2e-3
