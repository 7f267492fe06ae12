"""GPU parity: the fused GEGLU feed-forward launches (tb_ff_fwd / tb_ff_bwd, csrc/ff_fused.hip) against torch fp32 and against the
tb_gemm launches they replace (ff.net.0.proj + GEGLU epilogue, ff.net.2; their dgrads) -- diffusers BasicTransformerBlock.ff,
train_textboost.py:1063-1067 / :1108."""
import pytest
import torch
import torch.nn.functional as F

from parity import parity
from test_gpu_gemm import pack_geglu, rel_err

pytestmark = pytest.mark.gpu

C, INNER = 320, 1280


def _ops():
    from textboost_amd import ops, _lib
    return ops, _lib


def _weights(seed):
    g = torch.Generator().manual_seed(seed)
    w1 = (torch.randn(2 * INNER, C, generator=g) / C ** 0.5).half().cuda()       # ff.net.0.proj.weight (h rows, then g rows)
    b1 = (torch.randn(2 * INNER, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, INNER, generator=g) / INNER ** 0.5).half().cuda()       # ff.net.2.weight
    b2 = (torch.randn(C, generator=g) * 0.3).cuda()
    return w1, b1, w2, b2


def _ref_fwd(x, w1, b1, w2, b2, R):
    proj = x.float() @ w1.float().T + b1
    h, g = proj[:, :INNER].half().float(), proj[:, INNER:].half().float()     # the gate acts on the fp16-rounded projections
    u = (h * F.gelu(g)).half().float()
    y = u @ w2.float().T + b2
    return proj, y + (R.float() if R is not None else 0)


@pytest.mark.parametrize("M,with_r", [(128, True), (384, False), (4096, True)])
def test_ff_forward_vs_torch_and_vs_the_two_launch_path(M, with_r):
    ops, L = _ops()
    assert ops.ff_fused_ok(M, C, INNER) and not ops.ff_fused_ok(M + 64, C, INNER) and not ops.ff_fused_ok(M, 640, 2560)
    torch.manual_seed(M)
    w1, b1, w2, b2 = _weights(1)
    x = torch.randn(M, C, device="cuda").half()
    R = torch.randn(M, C, device="cuda").half() if with_r else None
    w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
    hg = torch.full((M, 2 * INNER + 16), 7.0, device="cuda", dtype=torch.float16)     # strided views: columns beyond the tensor stay untouched
    ybuf = torch.full((M, C + 8), 7.0, device="cuda", dtype=torch.float16)
    y = ops.ff_fwd(x, w1p, b1p, w2, b2, hg[:, :2 * INNER], ybuf[:, :C], R=R)
    proj, yref = _ref_fwd(x, w1, b1, w2, b2, R)
    parity("ff_fwd pre-gate projections (packed)", hg[:, :2 * INNER], pack_geglu(proj.T).T, 1e-3, 2e-3, ch_dim=1, ch_rel=2e-3)
    parity("ff_fwd output", y, yref, 2e-3, 3e-3, ch_dim=1, ch_rel=3e-3)
    assert (hg[:, 2 * INNER:] == 7).all() and (ybuf[:, C:] == 7).all()
    # the launches it replaces: the same arithmetic up to fp32 summation order (k-step width of the MFMA shape the dispatcher picked for this M)
    raw = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    gated = torch.empty(M, INNER, device="cuda", dtype=torch.float16)
    y2 = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.gemm(x, w1p, gated, bias=b1p, act=L.ACT_GEGLU, C2=raw)
    ops.gemm(gated, w2, y2, bias=b2, R=R)
    assert rel_err(raw, hg[:, :2 * INNER]) < 3e-4 and (raw.float() - hg[:, :2 * INNER].float()).abs().max().item() < 8e-3   # <= 1 fp16 ulp at |x| < 8
    assert rel_err(y, y2) < 3e-4


@pytest.mark.parametrize("M,with_r", [(128, False), (384, True), (4096, False)])
def test_ff_backward_vs_autograd_and_vs_the_two_launch_path(M, with_r):
    ops, L = _ops()
    torch.manual_seed(M + 1)
    w1, b1, w2, b2 = _weights(2)
    x = torch.randn(M, C, device="cuda").half()
    dy = torch.randn(M, C, device="cuda").half()
    R = torch.randn(M, C, device="cuda").half() if with_r else None
    w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
    hg = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    y = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_fwd(x, w1p, b1p, w2, b2, hg, y)
    w2d, w1d = w2.t().contiguous(), w1p.t().contiguous()      # the dgrad operands: ff.net.2.weight^T [inner, C], packed proj weight^T [C, 2 inner]
    dx = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_bwd(dy, w2d, w1d, hg, dx, R=R)
    # torch: the gradient of u @ w2^T with u = h * gelu(g) taken at the stored (fp16) projections, d(proj) rounded to fp16 like the operand of the
    # second product
    blocks = hg.float().reshape(M, INNER // 32, 2, 32)
    h = blocks[:, :, 0].reshape(M, INNER).requires_grad_(True)
    g = blocks[:, :, 1].reshape(M, INNER).requires_grad_(True)
    u = h * F.gelu(g)
    du = dy.float() @ w2.float()
    u.backward(du)
    dh, dg = h.grad.half().float(), g.grad.half().float()
    ref = dh @ w1[:INNER].float() + dg @ w1[INNER:].float() + (R.float() if R is not None else 0)
    parity("ff_bwd d(x)", dx, ref, 2e-3, 3e-3, ch_dim=1, ch_rel=3e-3)
    # the launches it replaces
    dproj = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    dx2 = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.gemm(dy, w2d, dproj, act=L.ACT_GEGLU_GRAD, C2=hg)
    ops.gemm(dproj, w1d, dx2, R=R)
    assert rel_err(dx, dx2) < 3e-4


@pytest.mark.parametrize("M", [256, 4096, 32768])
def test_ff_backward_with_the_layernorm_backward_in_its_epilogue(M):
    """round 5 (tb_ff_desc.ln_x / ln_stats / ln_gamma): the fused feed-forward backward applies norm3's LayerNorm backward to its accumulators and
    adds the residual gradient -- against autograd through LayerNorm -> ff and against the two launches it replaces (tb_ff_bwd + tb_layernorm_bwd)."""
    ops, L = _ops()
    torch.manual_seed(M + 7)
    w1, b1, w2, b2 = _weights(3)
    x = (torch.randn(M, C, device="cuda") * 0.8 + torch.randn(M, 1, device="cuda") * 0.5).half()     # the residual stream (LayerNorm input)
    gamma, beta = 1 + 0.4 * torch.randn(C, device="cuda"), 0.2 * torch.randn(C, device="cuda")
    l3 = torch.empty(M, C, device="cuda", dtype=torch.float16)
    st = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, l3, gamma, beta, st)
    dy = torch.randn(M, C, device="cuda").half()
    w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
    hg = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    y = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_fwd(l3, w1p, b1p, w2, b2, hg, y)
    w2d, w1d = w2.t().contiguous(), w1p.t().contiguous()
    dres = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_bwd(dy, w2d, w1d, hg, dres, R=dy, ln=(x, st, gamma))          # d(residual stream) = LN'(d l3) + dy
    # the launches it replaces
    dl3 = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_bwd(dy, w2d, w1d, hg, dl3)
    dres2 = torch.empty_like(dres)
    ops.layernorm_bwd(dl3, x, gamma, st, dres2, add=dy)
    assert rel_err(dres, dres2) < 6e-4
    # autograd from the stored projections on (as the test above), through the LayerNorm
    blocks = hg.float().reshape(M, INNER // 32, 2, 32)
    h = blocks[:, :, 0].reshape(M, INNER).requires_grad_(True)
    g = blocks[:, :, 1].reshape(M, INNER).requires_grad_(True)
    (h * F.gelu(g)).backward(dy.float() @ w2.float())
    dl3_ref = h.grad.half().float() @ w1[:INNER].float() + g.grad.half().float() @ w1[INNER:].float()
    xr = x.float().requires_grad_(True)
    F.layer_norm(xr, (C,), gamma, beta, 1e-5).backward(dl3_ref)
    parity("ff_bwd + LayerNorm backward", dres, xr.grad + dy.float(), 2e-3, 3e-3, ch_dim=1, ch_rel=3e-3)


@pytest.mark.parametrize("M,mode", [(128, 3), (384, 1), (384, 2), (4096, 3), (32768, 3)])
def test_ff_forward_with_its_row_local_neighbours_in_the_same_launch(M, mode):
    """round 6 (tb_ff_desc.pre_W / post_W): attn2.to_out + residual + norm3 in FRONT of the fused feed-forward (mode bit 1) and proj_out + the block
    input BEHIND it (bit 2) -- diffusers BasicTransformerBlock / Transformer2DModel, train_textboost.py:1063-1067 -- against torch fp32 on the same
    fp16-rounded intermediates and against the launches this replaces (tb_gemm with the LayerNorm epilogue, tb_ff_fwd, tb_gemm): the residual stream
    t2, the LayerNorm statistics, the stored pre-gate projections and the final output; strided views stay untouched outside their columns."""
    ops, L = _ops()
    torch.manual_seed(M + 11 * mode)
    w1, b1, w2, b2 = _weights(5)
    g = torch.Generator().manual_seed(9)
    wpre = (torch.randn(C, C, generator=g) / C ** 0.5).half().cuda()
    bpre = (torch.randn(C, generator=g) * 0.3).cuda()
    wpost = (torch.randn(C, C, generator=g) / C ** 0.5).half().cuda()
    bpost = (torch.randn(C, generator=g) * 0.3).cuda()
    gamma, beta = 1 + 0.4 * torch.randn(C, device="cuda"), 0.2 * torch.randn(C, device="cuda")
    o2 = torch.randn(M, C, device="cuda").half()                                                   # the cross-attention output
    t1 = (torch.randn(M, C, device="cuda") * 0.8 + torch.randn(M, 1, device="cuda") * 0.5).half()  # the residual stream in front of attn2.to_out
    xin = torch.randn(M, C, device="cuda").half()                                                  # the transformer's input (proj_out's residual)
    w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
    # ---- the launches this replaces
    t2_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    l3_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    st_old = torch.empty(M, 2, device="cuda")
    if ops.gemm_ln_ok(M, C, C):
        ops.gemm(o2, wpre, t2_old, bias=bpre, R=t1, ln_fwd=(gamma, beta, st_old, l3_old, 1e-5))
    else:
        ops.gemm(o2, wpre, t2_old, bias=bpre, R=t1)
        ops.layernorm_fwd(t2_old, l3_old, gamma, beta, st_old)
    hg_old = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    t3_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.ff_fwd(l3_old, w1p, b1p, w2, b2, hg_old, t3_old, R=t2_old)
    out_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.gemm(t3_old, wpost, out_old, bias=bpost, R=xin)
    # ---- one launch
    t2buf = torch.full((M, C + 8), 7.0, device="cuda", dtype=torch.float16)
    outbuf = torch.full((M, C + 16), 7.0, device="cuda", dtype=torch.float16)
    t2, out = t2buf[:, :C], outbuf[:, 8:8 + C]
    st = torch.zeros(M, 2, device="cuda")
    hg = torch.empty(M, 2 * INNER, device="cuda", dtype=torch.float16)
    t3 = torch.empty(M, C, device="cuda", dtype=torch.float16) if not (mode & 2) or M == 384 else None
    pre = (wpre, bpre, t1, t2, gamma, beta, st, 1e-5) if mode & 1 else None
    post = (wpost, bpost, xin, out) if mode & 2 else None
    if not (mode & 1):
        t2.copy_(t2_old)
    ops.ff_fwd(o2 if mode & 1 else l3_old, w1p, b1p, w2, b2, hg, t3, R=t2, pre=pre, post=post)
    # ---- torch fp32 on the fp16-rounded intermediates
    t2_ref = (o2.float() @ wpre.float().T + bpre + t1.float())
    t2h = t2_ref.half().float()
    l3_ref = F.layer_norm(t2h, (C,), gamma, beta, 1e-5).half().float()
    proj, t3_ref = _ref_fwd(l3_ref, w1, b1, w2, b2, t2h)
    out_ref = t3_ref.half().float() @ wpost.float().T + bpost + xin.float()
    if mode & 1:
        parity("chain: t2 = attn2.to_out + residual", t2, t2_ref, 1e-3, 2e-3, ch_dim=1, ch_rel=2e-3)
        assert torch.allclose(st[:, 0], t2h.mean(1), rtol=1e-4, atol=2e-4) and torch.allclose(st[:, 1], (t2h.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=3e-4, atol=0)
        assert rel_err(t2, t2_old) < 3e-4 and torch.allclose(st, st_old, rtol=5e-4, atol=3e-4)
        assert (t2buf[:, C:] == 7).all()
    parity("chain: pre-gate projections (packed)", hg, pack_geglu(proj.T).T, 2e-3, 4e-3, ch_dim=1, ch_rel=3e-3)
    assert rel_err(hg, hg_old) < 1.5e-3   # (a 1-ulp difference of a t2 element moves its whole LayerNorm row by an fp16 ulp)
    if t3 is not None:
        parity("chain: t3 = ff + residual", t3, t3_ref, 2e-3, 4e-3, ch_dim=1, ch_rel=3e-3)
    if mode & 2:
        parity("chain: proj_out + block input", out, out_ref, 2e-3, 4e-3, ch_dim=1, ch_rel=3e-3)
        assert rel_err(out, out_old) < 1e-3
        assert (outbuf[:, :8] == 7).all() and (outbuf[:, 8 + C:] == 7).all()


def test_ff_fused_rejects_what_it_does_not_cover():
    ops, L = _ops()
    d = L.FfDesc()
    d.M, d.C, d.inner = 100, C, INNER
    assert L.lib().tb_ff_fwd(d, None) == -22 and L.lib().tb_ff_bwd(d, None) == -22


@pytest.mark.parametrize("M,N2,with_r", [(128, 320, True), (384, 960, False), (4096, 320, False), (32768, 960, False), (32768, 320, True)])
def test_linear_layernorm_linear_in_one_launch(M, N2, with_r):
    """round 6 (tb_chain320, csrc/chain320.hip): proj_in -> norm1 -> attn1.qkv (N2 = 960) and attn1.to_out + residual -> norm2 -> attn2.to_q (N2 = 320)
    of a 64x64-map transformer block (diffusers Transformer2DModel / BasicTransformerBlock, train_textboost.py:1063-1067) as ONE launch -- against torch
    fp32 on the fp16-rounded residual stream and against the two launches it replaces (tb_gemm with the LayerNorm epilogue + tb_gemm); strided views."""
    ops, L = _ops()
    assert ops.chain320_ok(M, N2) and not ops.chain320_ok(M + 64, N2) and not ops.chain320_ok(M, 480)
    torch.manual_seed(M + N2)
    g = torch.Generator().manual_seed(21)
    w1 = (torch.randn(C, C, generator=g) / C ** 0.5).half().cuda()
    b1 = (torch.randn(C, generator=g) * 0.3).cuda()
    w2 = (torch.randn(N2, C, generator=g) / C ** 0.5).half().cuda()
    b2 = (torch.randn(N2, generator=g) * 0.3).cuda() if N2 == 320 else None        # (the qkv projection has no bias)
    gamma, beta = 1 + 0.4 * torch.randn(C, device="cuda"), 0.2 * torch.randn(C, device="cuda")
    x = torch.randn(M, C, device="cuda").half()
    R = (torch.randn(M, C, device="cuda") * 0.8 + torch.randn(M, 1, device="cuda") * 0.5).half() if with_r else None
    tbuf = torch.full((M, C + 8), 7.0, device="cuda", dtype=torch.float16)
    ybuf = torch.full((M, N2 + 16), 7.0, device="cuda", dtype=torch.float16)
    t, y = tbuf[:, :C], ybuf[:, 8:8 + N2]
    st = torch.zeros(M, 2, device="cuda")
    ops.chain320(x, w1, b1, R, t, gamma, beta, st, w2, b2, y)
    t_ref = x.float() @ w1.float().T + b1 + (R.float() if with_r else 0)
    th = t_ref.half().float()
    l_ref = F.layer_norm(th, (C,), gamma, beta, 1e-5).half().float()
    y_ref = l_ref @ w2.float().T + (b2 if b2 is not None else 0)
    parity("chain320: t = x W1^T + b (+ R)", t, t_ref, 1e-3, 2e-3, ch_dim=1, ch_rel=2e-3)
    assert torch.allclose(st[:, 0], th.mean(1), rtol=1e-4, atol=2e-4) and torch.allclose(st[:, 1], (th.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=3e-4, atol=0)
    parity("chain320: y = LN(t) W2^T + b", y, y_ref, 2e-3, 4e-3, ch_dim=1, ch_rel=3e-3)
    assert (tbuf[:, C:] == 7).all() and (ybuf[:, :8] == 7).all() and (ybuf[:, 8 + N2:] == 7).all()
    # the launches it replaces
    t_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    l_old = torch.empty(M, C, device="cuda", dtype=torch.float16)
    st_old = torch.empty(M, 2, device="cuda")
    if ops.gemm_ln_ok(M, C, C):
        ops.gemm(x, w1, t_old, bias=b1, R=R, ln_fwd=(gamma, beta, st_old, l_old, 1e-5))
    else:
        ops.gemm(x, w1, t_old, bias=b1, R=R)
        ops.layernorm_fwd(t_old, l_old, gamma, beta, st_old)
    y_old = torch.empty(M, N2, device="cuda", dtype=torch.float16)
    ops.gemm(l_old, w2, y_old, bias=b2)
    assert rel_err(t, t_old) < 3e-4 and torch.allclose(st, st_old, rtol=5e-4, atol=3e-4)
    assert rel_err(y, y_old) < 1.5e-3   # (a 1-ulp difference of a t element moves its whole LayerNorm row by an fp16 ulp)
    d = L.ChainDesc()
    d.M, d.N2 = 100, 320
    assert L.lib().tb_chain320(d, None) == -22
