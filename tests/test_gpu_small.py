"""GPU parity for the small streaming / text-encoder / optimizer kernels against torch references."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"


def _ops():
    from textboost_amd import ops, _lib
    return ops, _lib


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


def test_add_noise_and_velocity_and_timestep_embed():
    ops, L = _ops()
    from oracle import train_step as ts
    from oracle.unet_sd import timestep_embedding
    torch.manual_seed(0)
    B = 4
    x0 = torch.randn(B, 4, 16, 16, device=dev); n = torch.randn_like(x0)
    t = torch.tensor([0, 499, 998, 999], device=dev)
    acp = ts.alphas_cumprod().to(dev)
    noisy = torch.empty(B, 4, 16, 16, device=dev, dtype=torch.float16); vel = torch.empty_like(x0)
    ops.add_noise(x0, n, t, acp, noisy, vel)
    torch.testing.assert_close(noisy.float(), ts.add_noise(x0, n, t, acp), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(vel, ts.get_velocity(x0, n, t, acp), rtol=1e-5, atol=1e-5)
    out = torch.empty(B, 320, device=dev, dtype=torch.float16)
    ops.timestep_embed(t, out)
    torch.testing.assert_close(out.float(), timestep_embedding(t, 320), rtol=0, atol=2e-3)


def test_conv_in_out_boundary_kernels():
    ops, L = _ops()
    torch.manual_seed(1)
    B, H, W, C = 2, 12, 10, 64
    x = torch.randn(B, 4, H, W, device=dev).half()
    w = torch.randn(C, 4, 3, 3, device=dev) * 0.2; b = torch.randn(C, device=dev)
    wp = w.permute(2, 3, 1, 0).reshape(36, C).contiguous()           # [(tap*4+ci), Cout]
    out = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
    ops.conv4_to_nhwc(x, wp, b, out, B, H, W, C, sign=1)
    ref = F.conv2d(x.float(), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    assert rel_err(out, ref) < 2e-3
    # conv_out: NHWC [M, C] -> NCHW [B,4,H,W]
    wo = torch.randn(4, C, 3, 3, device=dev) * 0.1; bo = torch.randn(4, device=dev)
    h = torch.randn(B * H * W, C, device=dev).half()
    wop = wo.permute(0, 2, 3, 1).reshape(4, 9, C).contiguous()      # [co][tap][ci]
    pred = torch.empty(B, 4, H, W, device=dev, dtype=torch.float16)
    ops.conv_to4(h, wop, bo, pred, B, H, W, C)
    hn = h.float().view(B, H, W, C).permute(0, 3, 1, 2)
    ref = F.conv2d(hn, wo, bo, padding=1)
    assert rel_err(pred, ref) < 2e-3
    # conv_out dgrad: dpred fp32 NCHW -> dh NHWC, sign -1, weights [(tap*4+co), Cin]
    dpred = torch.randn(B, 4, H, W, device=dev)
    wdp = wo.permute(2, 3, 0, 1).reshape(36, C).contiguous()
    dh = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
    ops.conv4_to_nhwc(dpred, wdp, None, dh, B, H, W, C, sign=-1)
    ref = F.conv_transpose2d(dpred, wo, padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    assert rel_err(dh, ref) < 2e-3


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 32, 320), (1, 8, 16, 64), (2, 5, 48, 96)])
def test_conv_in_out_boundary_kernels_on_the_matrix_cores(B, H, W, C):
    """round 5: conv_in / conv_out / the conv_out input gradient as 16-pixel MFMA tiles (maps whose width is a multiple of 16, channels of 32;
    `tb_boundary_conv_set_variant`): against torch (the fp32 pack is consumed in 16 bits: inputs here are fp16-representable, as the weights of
    a model cast by unet.to(fp16) are) and against the fp32 VALU kernels they replace; image borders, the strided NHWC output, 3-channel input."""
    ops, L = _ops()
    torch.manual_seed(5)
    h16 = lambda t: t.half().float()   # noqa: E731
    x = torch.randn(B, 4, H, W, device=dev).half()
    w = h16(torch.randn(C, 4, 3, 3, device=dev) * 0.2); b = torch.randn(C, device=dev)
    wp = w.permute(2, 3, 1, 0).reshape(36, C).contiguous()
    wo = h16(torch.randn(4, C, 3, 3, device=dev) * 0.1); bo = torch.randn(4, device=dev)
    wop = wo.permute(0, 2, 3, 1).reshape(4, 9, C).contiguous()
    wdp = wo.permute(2, 3, 0, 1).reshape(36, C).contiguous()
    hbuf = torch.randn(B * H * W, C + 8, device=dev).half()
    h = hbuf[:, :C]
    dpred = torch.randn(B, 4, H, W, device=dev)
    x3 = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    w3 = h16(torch.randn(C, 3, 3, 3, device=dev) * 0.2)
    w3p = w3.permute(2, 3, 1, 0).reshape(27, C).contiguous()
    res = []
    old = L.lib().tb_boundary_conv_set_variant(1)
    try:
        for variant in (1, 0):
            L.lib().tb_boundary_conv_set_variant(variant)
            obuf = torch.full((B * H * W, C + 16), 3.0, device=dev, dtype=torch.float16)
            out = obuf[:, 8:8 + C]
            ops.conv4_to_nhwc(x, wp, b, out, B, H, W, C, sign=1)
            assert (obuf[:, :8] == 3).all() and (obuf[:, 8 + C:] == 3).all()
            pred = torch.empty(B, 4, H, W, device=dev, dtype=torch.float16)
            ops.conv_to4(h, wop, bo, pred, B, H, W, C)
            dh = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
            ops.conv4_to_nhwc(dpred, wdp, None, dh, B, H, W, C, sign=-1)
            o3 = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
            ops.convin_to_nhwc(x3, 3, w3p, b, o3, B, H, W, C)
            res.append((out.clone(), pred, dh, o3))
    finally:
        L.lib().tb_boundary_conv_set_variant(old)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])   # noqa: E731
    refs = (nhwc(F.conv2d(x.float(), w, b, padding=1)),
            F.conv2d(h.float().view(B, H, W, C).permute(0, 3, 1, 2), wo, bo, padding=1),
            nhwc(F.conv_transpose2d(dpred, wo, padding=1)),
            nhwc(F.conv2d(x3, w3, b, padding=1)))
    for k, (name, tol) in enumerate((("conv_in", 1e-3), ("conv_out", 1e-3), ("conv_out dgrad", 2e-3), ("conv_in rgb", 2e-3))):
        assert rel_err(res[0][k], refs[k]) < tol, (name, rel_err(res[0][k], refs[k]))
        assert rel_err(res[1][k], refs[k]) < tol, (name, "valu", rel_err(res[1][k], refs[k]))
        assert rel_err(res[0][k], res[1][k]) < tol, (name, "mfma vs valu", rel_err(res[0][k], res[1][k]))


def test_mse_and_kpl_losses():
    ops, L = _ops()
    torch.manual_seed(2)
    pred = torch.randn(2, 4, 16, 16, device=dev).half(); target = torch.randn(2, 4, 16, 16, device=dev)
    dpred = torch.empty_like(target); loss = torch.zeros(1, device=dev); ls = torch.tensor([1024.0], device=dev)
    ops.mse_loss(pred, target, dpred, loss, ls)
    pr = pred.float().requires_grad_(True)
    ref = F.mse_loss(pr, target); ref.backward()
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dpred, pr.grad * 1024, rtol=1e-5, atol=1e-6)
    # the metric's size (8 x 4 x 64 x 64 = 131072 elements) takes the two-stage multi-block reduction
    pred = torch.randn(8, 4, 64, 64, device=dev).half(); target = torch.randn(8, 4, 64, 64, device=dev)
    dpred = torch.empty_like(target)
    ops.mse_loss(pred, target, dpred, loss, ls)
    pr = pred.float().requires_grad_(True)
    ref = F.mse_loss(pr, target); ref.backward()
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dpred, pr.grad * 1024, rtol=1e-5, atol=1e-7)
    M, D = 154, 768
    h = torch.randn(M, D, device=dev); h0 = (h + 0.3 * torch.randn(M, D, device=dev)).half()
    dh = torch.empty_like(h); part = torch.empty(M, device=dev)
    ops.kpl_cos(h, h0, dh, part, loss, ls, 0.1)
    hr = h.clone().requires_grad_(True)
    ref = (1 - F.cosine_similarity(hr, h0.float(), dim=-1)).mean()
    (0.1 * 1024 * ref).backward()
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-4, atol=1e-6)
    assert rel_err(dh, hr.grad) < 1e-4


def test_geglu_bwd_pool_add_convert():
    ops, L = _ops()
    from tests.test_gpu_gemm import pack_geglu
    torch.manual_seed(3)
    M, inner = 96, 256
    proj = torch.randn(M, 2 * inner, device=dev).half()
    dout = torch.randn(M, inner, device=dev).half()
    raw = pack_geglu(proj.T.contiguous()).T.contiguous()
    dproj = torch.empty_like(raw)
    ops.geglu_bwd(dout, raw, dproj)
    pr = proj.float().requires_grad_(True)
    hh, gg = pr.chunk(2, dim=-1)
    (hh * F.gelu(gg)).backward(dout.float())
    assert rel_err(dproj, pack_geglu(pr.grad.T.contiguous()).T) < 2e-3
    B, H, W, C = 2, 5, 6, 64
    du = torch.randn(B * 4 * H * W, C, device=dev).half(); dx = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
    ops.pool2x2_sum(du, dx, B, H, W, C)
    ref = F.avg_pool2d(du.float().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2), 2) * 4
    assert rel_err(dx.view(B, H, W, C), ref.permute(0, 2, 3, 1)) < 2e-3
    # nearest x2 (diffusers Upsample2D's F.interpolate), bit exact, also into a column slice of a wider buffer
    xs = torch.randn(B * H * W, C, device=dev).half(); wide = torch.zeros(B * 4 * H * W, C + 16, device=dev, dtype=torch.float16)
    ops.upsample2x(xs, wide[:, 8:8 + C], B, H, W, C)
    refu = F.interpolate(xs.float().view(B, H, W, C).permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(wide[:, 8:8 + C].float().view(B, 2 * H, 2 * W, C), refu) and wide[:, :8].abs().max() == 0 and wide[:, 8 + C:].abs().max() == 0
    a = torch.randn(50, 128, device=dev).half(); bb = torch.randn(50, 256, device=dev).half(); o = torch.empty(50, 128, device=dev, dtype=torch.float16)
    ops.add_f16(a, bb[:, 64:192], o)
    torch.testing.assert_close(o.float(), a.float() + bb[:, 64:192].float(), rtol=2e-3, atol=2e-3)
    x32 = torch.randn(30, 77, device=dev); o16 = torch.empty(30, 77, device=dev, dtype=torch.float16)
    ops.convert(x32, o16, 2.0)
    torch.testing.assert_close(o16.float(), x32 * 2, rtol=2e-3, atol=2e-3)


def test_embed_pins_lora():
    ops, L = _ops()
    torch.manual_seed(4)
    B, T, D, V = 3, 77, 128, 200
    first = 190
    tok = torch.randn(V, D, device=dev); pos = torch.randn(T, D, device=dev)
    ids = torch.randint(0, V, (B, T), device=dev); ids[:, 0] = 5; ids[1, 1:] = 49407 % V
    eos = 49407 % V
    h = torch.empty(B * T, D, device=dev)
    ops.embed_fwd(ids.view(-1), tok, pos, h, T)
    torch.testing.assert_close(h.view(B, T, D), tok[ids] + pos[None])
    h16 = torch.empty(B * T, D, device=dev, dtype=torch.float16)
    ops.embed_fwd(ids.view(-1), tok.half(), pos.half(), h16, T)
    torch.testing.assert_close(h16.view(B, T, D), tok.half()[ids] + pos.half()[None])
    dh = torch.randn(B * T, D, device=dev)
    g = torch.zeros(V - first, D, device=dev)
    ops.embed_bwd(dh, ids.view(-1), g, first)
    ref = torch.zeros(V, D, device=dev).index_add_(0, ids.view(-1), dh)[first:]
    torch.testing.assert_close(g, ref, rtol=1e-5, atol=1e-5)
    null = torch.randn(T, D, device=dev)
    hh = h.clone()
    ops.pin_fwd(hh, ids.view(-1), null, B, T, use_fixed=True, eos_id=eos)
    ref = h.view(B, T, D).clone(); ref[1] = null; ref[:, 0] = null[0]
    torch.testing.assert_close(hh.view(B, T, D), ref)
    d2 = dh.clone()
    ops.pin_bwd(d2, ids.view(-1), B, T, use_fixed=True, eos_id=eos)
    ref = dh.view(B, T, D).clone(); ref[1] = 0; ref[:, 0] = 0
    torch.testing.assert_close(d2.view(B, T, D), ref)
    # LoRA pieces: P=3 adapters of rank r on a fused qkv projection
    M, K, Dm, r, P = B * T, 128, 128, 4, 3
    x = torch.randn(M, K, device=dev).half()
    A = torch.randn(P * r, K, device=dev) / r
    Bc = torch.randn(P * Dm, r, device=dev) * 0.1
    t = torch.zeros(M, 64, device=dev, dtype=torch.float16)
    ops.lora_down(x, A, t)
    assert rel_err(t[:, :P * r], x.float() @ A.half().float().T) < 2e-3 and t[:, P * r:].abs().max() == 0
    w2f = torch.empty(P * Dm, 64, device=dev, dtype=torch.float16); w2d = torch.empty(K, 64, device=dev, dtype=torch.float16)
    ops.lora_pack(A, Bc, w2f, w2d, Dm, K, r, P)
    ref = torch.zeros(P * Dm, 64, device=dev)
    for p_ in range(P):
        ref[p_ * Dm:(p_ + 1) * Dm, p_ * r:(p_ + 1) * r] = Bc[p_ * Dm:(p_ + 1) * Dm]
    torch.testing.assert_close(w2f.float(), ref.half().float())
    torch.testing.assert_close(w2d[:, :P * r].float(), A.T.half().float())
    assert w2d[:, P * r:].abs().max() == 0
    # stacked layers in one launch == per-layer launches
    A2l = torch.randn(2, P * r, K, device=dev) / r; B2l = torch.randn(2, P * Dm, r, device=dev) * 0.1
    w2f2 = torch.empty(2, P * Dm, 64, device=dev, dtype=torch.float16); w2d2 = torch.empty(2, K, 64, device=dev, dtype=torch.float16)
    ops.lora_pack(A2l, B2l, w2f2, w2d2, Dm, K, r, P, scaling=0.5, layers=2)
    for l in range(2):
        a1 = torch.empty(P * Dm, 64, device=dev, dtype=torch.float16); b1 = torch.empty(K, 64, device=dev, dtype=torch.float16)
        ops.lora_pack(A2l[l], B2l[l], a1, b1, Dm, K, r, P, scaling=0.5)
        assert torch.equal(a1, w2f2[l]) and torch.equal(b1, w2d2[l])
    # forward through the two-source GEMM equals x W^T + B(Ax)
    W = (torch.randn(P * Dm, K, device=dev) / 11).half()
    y = torch.empty(M, P * Dm, device=dev, dtype=torch.float16)
    ops.gemm(x, W, y, A2=t, W2=w2f)
    xr = x.float()
    lo = torch.cat([(xr @ A[p_ * r:(p_ + 1) * r].T) @ Bc[p_ * Dm:(p_ + 1) * Dm].T for p_ in range(P)], dim=1)
    assert rel_err(y, xr @ W.float().T + lo) < 3e-3
    # backward
    dY = torch.randn(M, P * Dm, device=dev).half()
    dt = torch.zeros(M, 64, device=dev, dtype=torch.float16)
    dA = torch.zeros_like(A); dB = torch.zeros_like(Bc)
    ops.lora_bwd(dY, x, t, Bc, dt, dA, dB, Dm, K, r, P)
    Ar = A.clone().requires_grad_(True); Br = Bc.clone().requires_grad_(True)
    lo = torch.cat([(xr @ Ar[p_ * r:(p_ + 1) * r].T) @ Br[p_ * Dm:(p_ + 1) * Dm].T for p_ in range(P)], dim=1)
    lo.backward(dY.float())
    assert rel_err(dA, Ar.grad) < 5e-3 and rel_err(dB, Br.grad) < 5e-3


def test_optimizer_tail_matches_torch():
    ops, L = _ops()
    from oracle import train_step as ts
    torch.manual_seed(5)
    n = 5000
    p = torch.randn(n, device=dev); pref = p.clone().cpu()
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    st = torch.zeros(L.ST_COUNT, device=dev); st[L.ST_LOSS_SCALE] = 1024.0
    ost = ts.AdamWState(lr=1e-3)
    sc = ts.GradScalerState(scale=1024.0, growth_interval=2)
    for it in range(4):
        g_true = torch.randn(n) * (3.0 if it != 2 else 1.0)
        g = (g_true * sc.scale).to(dev)
        if it == 2:
            g[7] = float("inf")
        ss = torch.zeros(2, device=dev)
        ops.sumsq(g, st[L.ST_SUMSQ_LORA:L.ST_SUMSQ_LORA + 1])
        st[L.ST_SUMSQ_EMB] = 0.0
        ops.scaler_update(st, max_norm=1.0, growth_interval=2)
        ops.adamw(p, g, m, v, 1e-3, st, L.ST_COEF_LORA)
        found = it == 2
        if not found:
            gg = [g_true.clone()]
            ts.clip_grad_norm(gg, 1.0)
            ts.adamw_step([pref], gg, ost)
        sc.update(found)
        assert abs(st[L.ST_LOSS_SCALE].item() - sc.scale) < 1e-3, (it, st[L.ST_LOSS_SCALE].item(), sc.scale)
        assert st[L.ST_FOUND_INF].item() == float(found)
        torch.testing.assert_close(p.cpu(), pref, rtol=2e-5, atol=2e-6)
    assert st[L.ST_STEP].item() == 3
    w = torch.randn(40, 768, device=dev); w0 = w.clone()
    ops.weight_decay(w, 1 - 1e-5, st)
    torch.testing.assert_close(w, w0 * (1 - 1e-5))
    rows = torch.randn(6, 768, device=dev) * torch.tensor([0.5, 1, 2, 0.1, 3, 1], device=dev)[:, None]
    r0 = rows.clone(); norms = torch.empty(6, device=dev)
    ops.renorm_rows(rows, 25.0, norms)
    vn = r0.norm(dim=-1, keepdim=True)
    torch.testing.assert_close(rows, torch.minimum(torch.full_like(vn, 25.0), vn) / vn * r0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(norms, vn[:, 0], rtol=1e-5, atol=1e-5)
    rn = torch.empty(40, device=dev)
    ops.row_norms(w, rn)
    torch.testing.assert_close(rn, w.norm(dim=-1), rtol=1e-5, atol=1e-5)


def test_gemm_extra_epilogues():
    ops, L = _ops()
    torch.manual_seed(6)
    M, N, K = 200, 256, 128
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / 11).half(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev); pre = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(A, W, out, bias=b, act=L.ACT_SILU)
    ref = F.silu(A.float() @ W.float().T + b)
    assert rel_err(out, ref) < 1e-3
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(A, W, o16, bias=b, act=L.ACT_QUICK_GELU, C2=pre)
    z = A.float() @ W.float().T + b
    assert rel_err(pre, z) < 1e-3 and rel_err(o16, z * torch.sigmoid(1.702 * z)) < 2e-3
    # dgrad with activation-gradient epilogue: dz = (dY W') * quick_gelu'(pre)
    dY = torch.randn(M, K, device=dev).half(); Wt = (torch.randn(N, K, device=dev) / 11).half()
    dz = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(dY, Wt, dz, act=L.ACT_QUICK_GELU_GRAD, C2=pre)
    pz = pre.float().requires_grad_(True)
    (pz * torch.sigmoid(1.702 * pz)).backward(dY.float() @ Wt.float().T)
    assert rel_err(dz, pz.grad) < 2e-3
    rb = torch.randn(4, 512, device=dev)
    o2 = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(A, W, o2, rowbias=rb[:, 100:100 + N], rows_per_group=50)
    ref = A.float() @ W.float().T + rb[:, 100:100 + N].repeat_interleave(50, 0)
    assert rel_err(o2, ref) < 2e-3


def test_gelu_epilogues_and_kpl_mse():
    ops, L = _ops()
    torch.manual_seed(7)
    M, N, K = 150, 256, 128
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / 11).half(); b = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16); pre = torch.empty_like(o16)
    ops.gemm(A, W, o16, bias=b, act=L.ACT_GELU, C2=pre)
    z = A.float() @ W.float().T + b
    assert rel_err(pre, z) < 1e-3 and rel_err(o16, F.gelu(z)) < 2e-3
    dY = torch.randn(M, K, device=dev).half(); Wt = (torch.randn(N, K, device=dev) / 11).half()
    dz = torch.empty(M, N, device=dev, dtype=torch.float16)
    ops.gemm(dY, Wt, dz, act=L.ACT_GELU_GRAD, C2=pre)
    pz = pre.float().requires_grad_(True)
    F.gelu(pz).backward(dY.float() @ Wt.float().T)
    assert rel_err(dz, pz.grad) < 2e-3
    Mh, D = 100, 768
    h = torch.randn(Mh, D, device=dev); h0 = (h + 0.3 * torch.randn(Mh, D, device=dev)).half()
    dh = torch.empty_like(h); part = torch.empty(Mh, device=dev); loss = torch.zeros(1, device=dev); ls = torch.tensor([512.0], device=dev)
    ops.kpl_mse(h, h0, dh, part, loss, ls, 0.1)
    hr = h.clone().requires_grad_(True)
    ref = F.mse_loss(hr, h0.float())
    (0.1 * 512 * ref).backward()
    torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-4, atol=1e-6)
    assert rel_err(dh, hr.grad) < 1e-5


def test_geglu_grad_epilogue_matches_separate_kernel():
    ops, L = _ops()
    from tests.test_gpu_gemm import pack_geglu
    torch.manual_seed(8)
    M, C = 200, 64
    inner = 4 * C
    dY = torch.randn(M, C, device=dev).half()
    Wd = (torch.randn(inner, C, device=dev) / 8).half()           # ff.net.2.weight^T : [4C, C]
    proj = torch.randn(M, 2 * inner, device=dev).half()
    raw = pack_geglu(proj.T.contiguous()).T.contiguous()
    dproj = torch.empty(M, 2 * inner, device=dev, dtype=torch.float16)
    ops.gemm(dY, Wd, dproj, act=L.ACT_GEGLU_GRAD, C2=raw)
    dgated = (dY.float() @ Wd.float().T)
    pr = proj.float().requires_grad_(True)
    hh, gg = pr.chunk(2, dim=-1)
    (hh * F.gelu(gg)).backward(dgated)
    assert rel_err(dproj, pack_geglu(pr.grad.T.contiguous()).T) < 3e-3


@pytest.mark.parametrize("r,Dm", [(2, 192), (4, 192), (8, 192), (4, 768), (8, 768), (4, 1024)])
def test_lora_backward_ranks(r, Dm):
    """tb_lora_bwd at the generic, rank-4 and rank-8 instantiations (the reference default r=4; SD2.1 config r=8) vs autograd; Dm = 768 / 1024
    take the dt slabs that hold every dY vector of a thread at once (CLIP-L / OpenCLIP-H widths)."""
    from textboost_amd import ops
    torch.manual_seed(r)
    M, K, P = 154, 256, 3
    x = torch.randn(M, K, device=dev).half()
    A = torch.randn(P * r, K, device=dev) / r
    Bc = torch.randn(P * Dm, r, device=dev) * 0.1
    t = torch.zeros(M, 64, device=dev, dtype=torch.float16)
    ops.lora_down(x, A, t)
    dY = torch.randn(M, P * Dm, device=dev).half()
    dt = torch.zeros(M, 64, device=dev, dtype=torch.float16)
    dA = torch.zeros_like(A); dB = torch.zeros_like(Bc)
    ops.lora_bwd(dY, x, t, Bc, dt, dA, dB, Dm, K, r, P, scaling=0.5)
    xr = x.float()
    Ar = A.clone().requires_grad_(True); Br = Bc.clone().requires_grad_(True)
    lo = 0.5 * torch.cat([(xr @ Ar[p_ * r:(p_ + 1) * r].T) @ Br[p_ * Dm:(p_ + 1) * Dm].T for p_ in range(P)], dim=1)
    lo.backward(dY.float())
    assert rel_err(dA, Ar.grad) < 5e-3 and rel_err(dB, Br.grad) < 5e-3
    ref_dt = 0.5 * torch.cat([dY.float()[:, p_ * Dm:(p_ + 1) * Dm] @ Bc[p_ * Dm:(p_ + 1) * Dm].half().float() for p_ in range(P)], dim=1)
    assert rel_err(dt[:, :P * r], ref_dt) < 3e-3 and dt[:, P * r:].abs().max() == 0


@pytest.mark.parametrize("r,Dm", [(4, 256), (8, 256), (4, 768), (8, 768), (4, 1024)])
def test_lora_backward_chain_is_bit_equal_to_per_set_launches(r, Dm):
    """tb_lora_bwd_chain (round 4): three adapter sets walked like the text encoder's layers -- each set's dA panels ride in the next set's dt / dB
    launch, the last set finishes its own -- against three tb_lora_bwd calls: same arithmetic, same summation order, bit for bit."""
    from textboost_amd import ops
    torch.manual_seed(10 + r)
    M, K, P, n = 154, 256, 3, 3      # (Dm = 768 / 1024: the dt slabs hold all of a thread's dY vectors at once)
    xs = [torch.randn(M, K, device=dev).half() for _ in range(n)]
    As = [torch.randn(P * r, K, device=dev) / r for _ in range(n)]
    Bs = [torch.randn(P * Dm, r, device=dev) * 0.1 for _ in range(n)]
    dYs = [torch.randn(M, P * Dm, device=dev).half() for _ in range(n)]
    ts = []
    for x, A in zip(xs, As):
        t = torch.zeros(M, 64, device=dev, dtype=torch.float16)
        ops.lora_down(x, A, t)
        ts.append(t)
    ref = []
    for i in range(n):
        dt = torch.zeros(M, 64, device=dev, dtype=torch.float16)
        dA = torch.full_like(As[i], 0.25); dB = torch.full_like(Bs[i], -0.5)      # (+=: the gradients accumulate)
        ops.lora_bwd(dYs[i], xs[i], ts[i], Bs[i], dt, dA, dB, Dm, K, r, P, scaling=0.5)
        ref.append((dt, dA, dB))
    dts = [torch.zeros(M, 64, device=dev, dtype=torch.float16) for _ in range(2)]
    dAs = [torch.full_like(A, 0.25) for A in As]; dBs = [torch.full_like(B, -0.5) for B in Bs]
    pending, seen_dt = None, []
    for i in reversed(range(n)):
        dt = dts[i & 1]
        pending = ops.lora_bwd(dYs[i], xs[i], ts[i], Bs[i], dt, dAs[i], dBs[i], Dm, K, r, P, scaling=0.5, pending=pending, defer_da=i > 0)
        seen_dt.append((i, dt.clone()))
    assert pending is None
    for i, dt in seen_dt:
        assert torch.equal(dt, ref[i][0])
    for i in range(n):
        assert torch.equal(dAs[i], ref[i][1]) and torch.equal(dBs[i], ref[i][2]), i
    # a pending set that shares this link's dt buffer is refused (its dt would be overwritten under the reader)
    with pytest.raises(RuntimeError):
        ops.lora_bwd(dYs[0], xs[0], ts[0], Bs[0], dts[0], dAs[0], dBs[0], Dm, K, r, P, pending=(xs[1], dts[0].clone(), dAs[1])[:1] + (dts[0], dAs[1]))


def test_lr_multiplier_slot_scales_every_group():
    """state[TB_ST_LR_MULT] = lambda - 1 (lr_scheduler, :911-916/:1135): AdamW and the decay-only rows see lr * lambda."""
    from textboost_amd import _lib as L, ops
    torch.manual_seed(0)
    n = 4096
    p0 = torch.randn(n, device=dev); g = torch.randn(n, device=dev)
    outs = []
    for lr, lam in ((1e-3, 0.25), (0.25e-3, 1.0)):
        p = p0.clone(); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
        st = torch.zeros(L.ST_COUNT, device=dev); st[L.ST_LOSS_SCALE] = 1.0
        ops.sumsq(g, st[L.ST_SUMSQ_LORA:L.ST_SUMSQ_LORA + 1])
        ops.scaler_update(st, max_norm=1e9, growth_interval=2000)
        st[L.ST_LR_MULT] = lam - 1.0
        ops.adamw(p, g, m, v, lr, st, L.ST_COEF_LORA)
        w = p0.clone().view(4, -1).contiguous()
        ops.weight_decay(w.view(-1), 1 - lr * 1e-2, st)
        outs.append((p, w))
    torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-6, atol=1e-8)
    assert not torch.equal(outs[0][0], p0)
