"""GPU parity: tb_gemm (linear / conv3x3 gathers / epilogues) against plain torch fp32 references."""
import pytest
import torch
import torch.nn.functional as F

from parity import parity

pytestmark = pytest.mark.gpu


def _ops():
    from textboost_amd import ops, _lib
    return ops, _lib


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (616, 768, 768), (300, 320, 640), (1024, 2304, 832), (130, 70, 128)])
def test_linear_plain(M, N, K):
    ops, L = _ops()
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    ops.gemm(A, W, out, bias=bias)
    ref = A.float() @ W.float().T + bias
    assert rel_err(out, ref) < 2e-3
    # asymmetric / transpose-detecting: max abs error too
    assert (out.float() - ref).abs().max().item() < 2e-2


def test_linear_strided_two_source_residual_fp32_out():
    ops, L = _ops()
    torch.manual_seed(1)
    M, N, K1, K2 = 616, 2304, 768, 64
    Abuf = torch.randn(M, K1 + 40, device="cuda").half()   # lda > K
    A = Abuf[:, :K1]
    A2 = torch.randn(M, K2, device="cuda").half()
    W = (torch.randn(N, K1, device="cuda") / 30).half()
    W2 = (torch.randn(N, K2, device="cuda") / 30).half()
    R = torch.randn(M, N, device="cuda")
    Cbuf = torch.zeros(M, N + 16, device="cuda")
    out = Cbuf[:, 8:8 + N]
    ops.gemm(A, W, out, A2=A2, W2=W2, R=R, alpha=0.5, act=L.ACT_QUICK_GELU)
    pre = 0.5 * (A.float() @ W.float().T + A2.float() @ W2.float().T) + R
    ref = pre * torch.sigmoid(1.702 * pre)
    assert rel_err(out, ref) < 1e-3
    assert Cbuf[:, :8].abs().max() == 0 and Cbuf[:, 8 + N:].abs().max() == 0


def test_linear_rowbias_f16_residual():
    ops, L = _ops()
    torch.manual_seed(2)
    B, HW, N, K = 3, 100, 320, 320
    A = torch.randn(B * HW, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / 18).half()
    rb = torch.randn(B, N, device="cuda")
    R = torch.randn(B * HW, N, device="cuda").half()
    out = torch.empty(B * HW, N, device="cuda", dtype=torch.float16)
    ops.gemm(A, W, out, rowbias=rb, rows_per_group=HW, R=R)
    ref = A.float() @ W.float().T + rb.repeat_interleave(HW, 0) + R.float()
    assert rel_err(out, ref) < 2e-3


def pack_geglu(w):
    """[2*inner, ...] rows (h rows then g rows) -> 32-row interleaved blocks [h0..31 | g0..31 | h32..63 | ...]."""
    inner = w.shape[0] // 2
    h, g = w[:inner], w[inner:]
    hs = h.reshape(inner // 32, 32, *w.shape[1:])
    gs = g.reshape(inner // 32, 32, *w.shape[1:])
    return torch.stack([hs, gs], dim=1).reshape(w.shape)


def test_geglu():
    ops, L = _ops()
    torch.manual_seed(3)
    M, C = 384, 320
    A = torch.randn(M, C, device="cuda").half()
    W = (torch.randn(8 * C, C, device="cuda") / 18).half()
    b = torch.randn(8 * C, device="cuda")
    out = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    raw = torch.empty(M, 8 * C, device="cuda", dtype=torch.float16)
    ops.gemm(A, pack_geglu(W), out, bias=pack_geglu(b), act=L.ACT_GEGLU, C2=raw)
    proj = A.float() @ W.float().T + b
    h, g = proj.chunk(2, dim=-1)
    ref = h * F.gelu(g)
    assert rel_err(out, ref) < 3e-3
    assert rel_err(raw, pack_geglu(proj.T).T) < 2e-3


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def pack_conv_w(w):  # [Cout, Cin, 3, 3] -> [Cout, 9*Cin] with k = (ky*3+kx)*Cin + ci
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_conv_w_dgrad(w):  # -> [Cin, 9*Cout] with k = (ky*3+kx)*Cout + co
    return w.permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous()


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 128, 16, 16), (1, 320, 320, 24, 20), (2, 128, 64, 7, 9)])
def test_conv3x3_fwd_and_dgrad(B, Cin, Cout, H, W):
    ops, L = _ops()
    torch.manual_seed(4)
    x = torch.randn(B, Cin, H, W, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    xn = nhwc(x).view(B * H * W, Cin)
    out = torch.empty(B * H * W, Cout, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cin, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=0)
    ops.gemm(xn, pack_conv_w(w), out, bias=bias, conv=geo)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    assert rel_err(out.view(B, H, W, Cout), nhwc(ref)) < 2e-3
    # dgrad: dX = conv_transpose(dY, w) == same kernel with sign=-1 on [Cin][tap][Cout] weights
    dy = torch.randn(B, Cout, H, W, device="cuda").half()
    dx = torch.empty(B * H * W, Cin, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cout, Hout=H, Wout=W, stride=1, sign=-1, upsample=0, transposed=0)
    ops.gemm(nhwc(dy).view(B * H * W, Cout), pack_conv_w_dgrad(w), dx, conv=geo)
    ref = F.conv_transpose2d(dy.float(), w.float(), padding=1)
    assert rel_err(dx.view(B, H, W, Cin), nhwc(ref)) < 2e-3


def test_conv3x3_stride2_fwd_and_transposed_dgrad():
    ops, L = _ops()
    torch.manual_seed(5)
    B, Ci, Co, H, W = 2, 64, 64, 16, 12
    x = torch.randn(B, Ci, H, W, device="cuda").half()
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") / 24).half()
    Ho, Wo = H // 2, W // 2
    out = torch.empty(B * Ho * Wo, Co, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=Ci, Hout=Ho, Wout=Wo, stride=2, sign=1, upsample=0, transposed=0)
    ops.gemm(nhwc(x).view(-1, Ci), pack_conv_w(w), out, conv=geo)
    ref = F.conv2d(x.float(), w.float(), stride=2, padding=1)
    assert rel_err(out.view(B, Ho, Wo, Co), nhwc(ref)) < 2e-3
    dy = torch.randn(B, Co, Ho, Wo, device="cuda").half()
    dx = torch.empty(B * H * W, Ci, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=Ho, Win=Wo, Cin=Co, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=1)
    ops.gemm(nhwc(dy).view(-1, Co), pack_conv_w_dgrad(w), dx, conv=geo)
    ref = F.conv_transpose2d(dy.float(), w.float(), stride=2, padding=1, output_padding=1)
    assert rel_err(dx.view(B, H, W, Ci), nhwc(ref)) < 2e-3


def test_conv3x3_upsample_folded():
    ops, L = _ops()
    torch.manual_seed(6)
    B, Cc, H, W = 2, 128, 8, 6
    x = torch.randn(B, Cc, H, W, device="cuda").half()
    w = (torch.randn(Cc, Cc, 3, 3, device="cuda") / 34).half()
    out = torch.empty(B * 4 * H * W, Cc, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cc, Hout=2 * H, Wout=2 * W, stride=1, sign=1, upsample=1, transposed=0)
    ops.gemm(nhwc(x).view(-1, Cc), pack_conv_w(w), out, conv=geo)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), padding=1)
    assert rel_err(out.view(B, 2 * H, 2 * W, Cc), nhwc(ref)) < 2e-3


def test_gemm_rejects_bad_args():
    ops, L = _ops()
    A = torch.zeros(8, 100, device="cuda", dtype=torch.float16)  # K not multiple of 64
    W = torch.zeros(8, 100, device="cuda", dtype=torch.float16)
    out = torch.zeros(8, 8, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.gemm(A, W, out)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(8, 64, 128, 64, 64), (8, 128, 320, 64, 64), (8, 64, 640, 32, 32), (8, 64, 2048, 16, 16),
                                            (2, 192, 1280, 64, 64), (2, 64, 128, 48, 128), (1, 128, 64, 24, 256), (1, 64, 128, 8, 512), (4, 64, 128, 96, 96),
                                            (8, 128, 64, 48, 48), (16, 64, 64, 32, 24), (8, 640, 1280, 8, 8), (4, 1280, 1280, 8, 8)])
def test_conv3x3_halo_kernel_fwd_and_dgrad(B, Cin, Cout, H, W):
    """shapes that take the LDS-halo path (whole image rows per 128-pixel tile, 128-pixel segments of rows 128 / 256 / 512 wide
    as in the VAE encoder, or -- round 4 -- two whole 8x8 images per tile, each with its own zero border; >= 200 tiles incl. k-slices),
    checked against F.conv2d and against the per-tap gather kernel (halo path switched off)."""
    ops, L = _ops()
    torch.manual_seed(7)
    x = torch.randn(B, Cin, H, W, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(B * H * W, Cout, device="cuda").half()
    xn = nhwc(x).view(B * H * W, Cin)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cin, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=0)
    outs = []
    for halo in (3, 0):
        L.lib().tb_gemm_set_variant(7000 + halo)
        out = torch.empty(B * H * W, Cout, device="cuda", dtype=torch.float16)
        ops.gemm(xn, pack_conv_w(w), out, bias=bias, R=res, conv=geo)
        outs.append(out)
        if halo and H * W < 128:   # the two-images-per-tile case must really run the halo kernel
            cfg = (__import__("ctypes").c_int * 5)()
            L.lib().tb_gemm_last_config(cfg)
            assert cfg[2] == 2, "the shape must take conv_halo_kernel"
    L.lib().tb_gemm_set_variant(7003)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * W, Cout) + res.float()
    assert rel_err(outs[0], ref) < 2e-3
    assert rel_err(outs[0], outs[1]) < 1e-3   # same products, different fp32 summation order (chunk-outer vs tap-outer)
    dy = torch.randn(B, Cout, H, W, device="cuda").half()
    geo = dict(B=B, Hin=H, Win=W, Cin=Cout, Hout=H, Wout=W, stride=1, sign=-1, upsample=0, transposed=0)
    dx = torch.empty(B * H * W, Cin, device="cuda", dtype=torch.float16)
    ops.gemm(nhwc(dy).view(B * H * W, Cout), pack_conv_w_dgrad(w), dx, conv=geo)
    refd = nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1))
    assert rel_err(dx.view(B, H, W, Cin), refd) < 2e-3


@pytest.mark.parametrize("M,C", [(32768, 320), (8192, 640)])
def test_geglu_forward_and_backward_epilogues_on_the_wide_tile_kernel(M, C):
    """ff.net.0.proj + GEGLU (N = 8C a multiple of 640) and the GEGLU backward fused into the ff.net.2 dgrad GEMM (N = 4C a multiple of
    320) at row counts that select gemm8_kernel's 128 x 320 tiles: against torch and against the 4-wave kernels' results."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(5)
    A = torch.randn(M, C, device="cuda").half()
    W = (torch.randn(8 * C, C, device="cuda") / C ** 0.5).half()
    b = torch.randn(8 * C, device="cuda")
    outs = []
    default_bits = L.lib().tb_gemm8_set(7)
    for bits in (7, 0):
        L.lib().tb_gemm8_set(bits)
        out = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
        raw = torch.empty(M, 8 * C, device="cuda", dtype=torch.float16)
        ops.gemm(A, pack_geglu(W), out, bias=pack_geglu(b), act=L.ACT_GEGLU, C2=raw)
        used = L.lib().tb_gemm8_last(None)
        assert bool(used) == (bits == 7)
        outs.append((out, raw))
    L.lib().tb_gemm8_set(default_bits)
    proj = A.float() @ W.float().T + b
    h, g = proj.chunk(2, dim=-1)
    ref = h * F.gelu(g)
    for out, raw in outs:
        assert rel_err(out, ref) < 3e-3 and rel_err(raw, pack_geglu(proj.T).T) < 2e-3
    assert torch.equal(outs[0][1], outs[1][1]) or rel_err(outs[0][1], outs[1][1]) < 1e-3
    # backward: d(gated) = dY @ Wd^T, then d(proj) through the gate
    dY = torch.randn(M, C, device="cuda").half()
    Wd = (torch.randn(4 * C, C, device="cuda") / C ** 0.5).half()
    raw = outs[0][1]
    res = []
    for bits in (7, 0):
        L.lib().tb_gemm8_set(bits)
        dproj = torch.empty(M, 8 * C, device="cuda", dtype=torch.float16)
        ops.gemm(dY, Wd, dproj, act=L.ACT_GEGLU_GRAD, C2=raw)
        assert bool(L.lib().tb_gemm8_last(None)) == (bits == 7)
        res.append(dproj)
    L.lib().tb_gemm8_set(default_bits)
    dgated = dY.float() @ Wd.float().T
    unpack = lambda t: torch.cat([t.view(M, -1, 2, 32)[:, :, 0].reshape(M, -1), t.view(M, -1, 2, 32)[:, :, 1].reshape(M, -1)], dim=1)
    pr = unpack(raw.float()).requires_grad_(True)
    hh, gg = pr.chunk(2, dim=-1)
    (hh * F.gelu(gg)).backward(dgated)
    for dproj in res:
        assert rel_err(unpack(dproj.float()), pr.grad) < 3e-3


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(8, 128, 256, 64, 64), (2, 128, 128, 256, 256), (1, 256, 512, 128, 512), (8, 512, 512, 32, 64)])
def test_conv3x3_wide_tile_128_channels(B, Cin, Cout, H, W):
    """the 256-pixel x 128-channel tile of gemm8_kernel (round 4: the VAE's channel counts, which the 80-wide wave tiles do not divide), forward
    with bias + residual and dgrad, against F.conv2d and against the 4-wave LDS-halo kernel."""
    ops, L = _ops()
    torch.manual_seed(12)
    x = torch.randn(B, Cin, H, W, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(B * H * W, Cout, device="cuda").half()
    geo = dict(B=B, Hin=H, Win=W, Cin=Cin, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=0)
    default_bits = L.lib().tb_gemm8_set(39)
    outs = []
    for bits in (default_bits, default_bits | 1024):
        L.lib().tb_gemm8_set(bits)
        out = torch.empty(B * H * W, Cout, device="cuda", dtype=torch.float16)
        ops.gemm(nhwc(x).view(B * H * W, Cin), pack_conv_w(w), out, bias=bias, R=res, conv=geo)
        assert bool(L.lib().tb_gemm8_last(None)) == (bits == default_bits)
        outs.append(out)
    L.lib().tb_gemm8_set(default_bits)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * W, Cout) + res.float()
    parity("wide-tile conv, 128-channel tiles", outs[0], ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
    assert rel_err(outs[0], outs[1]) < 1e-3
    dy = torch.randn(B, Cout, H, W, device="cuda").half()
    dx = torch.empty(B * H * W, Cin, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cout, Hout=H, Wout=W, stride=1, sign=-1, upsample=0, transposed=0)
    ops.gemm(nhwc(dy).view(B * H * W, Cout), pack_conv_w_dgrad(w), dx, conv=geo)
    parity("wide-tile conv dgrad, 128-channel tiles", dx, nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1)).view(B * H * W, Cin),
           rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)


@pytest.mark.parametrize("B,Cin,Cout,H,rowbias", [(8, 640, 640, 32, True), (8, 1280, 640, 32, False), (8, 320, 640, 32, True), (8, 1920, 640, 32, False)])
def test_conv3x3_256x80_tile_on_the_four_slot_ring(B, Cin, Cout, H, rowbias):
    """gemm8_kernel<8, 1, 2, 5, true, 4> (round 5, DMACH): the 32x32-map convolutions on a 4-slot weight ring with the global -> LDS pieces issued
    between the MFMAs of the compute phase and run-time ring slots -- forward with bias, the time-embedding row bias (its group now comes from the tile's
    image index) and residual, and the dgrad, against F.conv2d and BIT-equal to round 4's 3-slot loop (tb_gemm8_set bit 32768: same products, same
    summation order)."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(21)
    x = torch.randn(B, Cin, H, H, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    rb = torch.randn(B, Cout, device="cuda") if rowbias else None
    res = torch.randn(B * H * H, Cout, device="cuda").half()
    xn = nhwc(x).view(B * H * H, Cin)
    geo = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    default_bits = L.lib().tb_gemm8_set(39)
    outs, dxs = [], []
    dy = torch.randn(B, Cout, H, H, device="cuda").half()
    geod = dict(B=B, Hin=H, Win=H, Cin=Cout, Hout=H, Wout=H, stride=1, sign=-1, upsample=0, transposed=0)
    try:
        for bits, ns in ((39, 4), (39 | 32768, 3)):
            L.lib().tb_gemm8_set(bits)
            out = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.float16)
            ops.gemm(xn, pack_conv_w(w), out, bias=bias, R=res, conv=geo, rowbias=rb, rows_per_group=H * H if rowbias else 0)
            last = (ctypes.c_int * 6)()
            assert L.lib().tb_gemm8_last(last) and list(last) == [8, 1, 2, 5, 1, ns], list(last)
            outs.append(out)
            dx = torch.empty(B * H * H, Cin, device="cuda", dtype=torch.float16)
            ops.gemm(nhwc(dy).view(B * H * H, Cout), pack_conv_w_dgrad(w), dx, conv=geod)
            dxs.append(dx)
    finally:
        L.lib().tb_gemm8_set(default_bits)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * H, Cout) + res.float()
    if rowbias:
        ref = ref + rb.repeat_interleave(H * H, dim=0)
    parity("256 x 80 conv tile, 4-slot ring", outs[0], ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(dxs[0], dxs[1])
    parity("256 x 80 conv tile dgrad, 4-slot ring", dxs[0], nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1)).view(B * H * H, Cin),
           rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)


@pytest.mark.parametrize("B,Cin,Cout,H", [(8, 640, 640, 16), (8, 1280, 1280, 16), (8, 2560, 1280, 16), (4, 1280, 640, 32)])
def test_conv3x3_wide_tile_split_k(B, Cin, Cout, H):
    """16x16 / small-batch maps give too few 256 x 160 tiles for the chip: gemm8_kernel splits the channel chunks over S workgroups per tile
    (fp32 partials + splitk_reduce_kernel).  Forward with bias, SiLU-free residual epilogue and dgrad, against F.conv2d and the 4-wave path."""
    ops, L = _ops()
    torch.manual_seed(11)
    x = torch.randn(B, Cin, H, H, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(B * H * H, Cout, device="cuda").half()
    xn = nhwc(x).view(B * H * H, Cin)
    geo = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    default_bits = L.lib().tb_gemm8_set(39)
    outs = []
    for bits in (39, 7, 0):
        L.lib().tb_gemm8_set(bits)
        out = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.float16)
        ops.gemm(xn, pack_conv_w(w), out, bias=bias, R=res, conv=geo)
        if bits == 39:
            assert L.lib().tb_gemm8_last(None), "the split-K wide-tile path did not take this shape"
        outs.append(out)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * H, Cout) + res.float()
    for o in outs:
        assert rel_err(o, ref) < 2e-3
    assert rel_err(outs[0], outs[2]) < 1e-3
    dy = torch.randn(B, Cout, H, H, device="cuda").half()
    geo = dict(B=B, Hin=H, Win=H, Cin=Cout, Hout=H, Wout=H, stride=1, sign=-1, upsample=0, transposed=0)
    L.lib().tb_gemm8_set(39)
    dx = torch.empty(B * H * H, Cin, device="cuda", dtype=torch.float16)
    ops.gemm(nhwc(dy).view(B * H * H, Cout), pack_conv_w_dgrad(w), dx, conv=geo)
    L.lib().tb_gemm8_set(default_bits)
    refd = nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1))
    assert rel_err(dx.view(B, H, H, Cin), refd) < 2e-3


@pytest.mark.parametrize("M,K,res", [(32768, 320, True), (32768, 960, False), (16384, 320, True), (25600, 640, True)])
def test_layernorm_fused_into_the_linear_epilogue_forward(M, K, res):
    """TB_ACT_LN_FWD (round 3): the Linear that produces the residual stream also writes LayerNorm(row) and its statistics -- against
    torch.nn.functional.layer_norm of the fp16 row it stored, and against the separate tb_layernorm_fwd launch.  128 x 320 tiles (M / 128 >= 200)
    and 64 x 320 tiles; with and without residual; strided outputs."""
    from parity import parity
    ops, L = _ops()
    torch.manual_seed(3)
    N = 320
    assert ops.gemm_ln_ok(M, N, K) and not ops.gemm_ln_ok(4096, N, K) and not ops.gemm_ln_ok(M, 640, K)
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    bias = torch.randn(N, device="cuda")
    R = (torch.randn(M, N, device="cuda") * 2 + 0.5).half() if res else None
    gamma, beta = torch.randn(N, device="cuda") * 0.5 + 1, torch.randn(N, device="cuda") * 0.1
    tbuf = torch.zeros(M, N + 8, device="cuda", dtype=torch.float16)
    t, y = tbuf[:, :N], torch.empty(M, N, device="cuda", dtype=torch.float16)
    stats = torch.empty(M, 2, device="cuda")
    ops.gemm(A, W, t, bias=bias, R=R, ln_fwd=(gamma, beta, stats, y, 1e-5))
    # the Linear itself
    t_ref = torch.empty(M, N, device="cuda", dtype=torch.float16)
    ops.gemm(A, W, t_ref, bias=bias, R=R)
    assert torch.equal(t, t_ref) and tbuf[:, N:].abs().max() == 0
    # statistics and the normalised row, from the stored fp16 row
    tf = t.float()
    mean, var = tf.mean(1), tf.var(1, unbiased=False)
    torch.testing.assert_close(stats[:, 0], mean, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[:, 1], torch.rsqrt(var + 1e-5), rtol=1e-5, atol=0)
    parity("fused LN forward vs torch", y, F.layer_norm(tf, (N,), gamma, beta, 1e-5), rel=6e-4, maxabs=1e-3, ch_dim=1, ch_rel=1e-3)
    y2, st2 = torch.empty_like(y), torch.empty_like(stats)
    ops.layernorm_fwd(t_ref, y2, gamma, beta, st2)
    torch.testing.assert_close(stats, st2, rtol=1e-5, atol=1e-5)
    assert (y.float() - y2.float()).abs().max().item() <= 2e-3 * y2.float().abs().max().item()   # one fp16 ulp where a value sits on a rounding edge


@pytest.mark.parametrize("M,K,res", [(32768, 320, True), (32768, 2560, True), (16384, 960, False)])
def test_layernorm_backward_fused_into_the_dgrad_epilogue(M, K, res):
    """TB_ACT_LN_BWD: dx = LN'(dy = A W^T) + add on the GEMM's accumulators, against torch autograd of layer_norm on the same operands."""
    from parity import parity
    ops, L = _ops()
    torch.manual_seed(4)
    N = 320
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    x = (torch.randn(M, N, device="cuda") * 1.5 + 0.3).half()
    add = torch.randn(M, N, device="cuda").half() if res else None
    gamma, beta = torch.randn(N, device="cuda") * 0.5 + 1, torch.zeros(N, device="cuda")
    y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, y, gamma, beta, stats)
    dx = torch.empty(M, N, device="cuda", dtype=torch.float16)
    ops.gemm(A, W, dx, R=add, ln_bwd=(gamma, stats, x))
    xr = x.float().requires_grad_(True)
    dy = A.float() @ W.float().T
    F.layer_norm(xr, (N,), gamma, beta, 1e-5).backward(dy)
    ref = xr.grad + (add.float() if res else 0)
    parity("fused LN backward vs torch autograd", dx, ref, rel=6e-4, maxabs=1.5e-3, ch_dim=1, ch_rel=1e-3)
    # and the two-launch path it replaces (dy rounded to fp16 in between)
    dy16, dx2 = torch.empty(M, N, device="cuda", dtype=torch.float16), torch.empty_like(dx)
    ops.gemm(A, W, dy16)
    ops.layernorm_bwd(dy16, x, gamma, stats, dx2, add=add)
    parity("fused LN backward vs gemm + tb_layernorm_bwd", dx, dx2, rel=1e-3, maxabs=3e-3)


def test_layernorm_epilogue_is_refused_where_no_tile_spans_the_row():
    ops, L = _ops()
    A, W = torch.randn(4096, 320, device="cuda").half(), torch.randn(320, 320, device="cuda").half()
    out, y, st = torch.empty(4096, 320, device="cuda", dtype=torch.float16), torch.empty(4096, 320, device="cuda", dtype=torch.float16), torch.empty(4096, 2, device="cuda")
    g = torch.ones(320, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm(A, W, out, ln_fwd=(g, g, st, y, 1e-5))


@pytest.mark.parametrize("B,Co,Ci,H,W,res", [(2, 64, 64, 16, 16, False), (2, 128, 192, 32, 16, True), (8, 320, 320, 64, 64, False), (8, 1280, 1280, 16, 16, True)])
def test_transposed_dgrad_phase_ordered_rows(B, Co, Ci, H, W, res):
    """stride-2 dgrad with the rows walked parity class by parity class (tb_gemm switches to it when M / 4 is a whole number of 128-row tiles):
    against conv_transpose2d and against the map-order gather it replaces (skipped taps only ever added exact zeros: bit-equal unless the launch
    is split over K, where the slice boundaries move with the shorter tap lists)."""
    from parity import parity
    ops, L = _ops()
    torch.manual_seed(7)
    Ho, Wo = H // 2, W // 2
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") / (3 * Co ** 0.5)).half()
    dy = torch.randn(B, Co, Ho, Wo, device="cuda").half()
    R = torch.randn(B * H * W, Ci, device="cuda").half() if res else None
    geo = dict(B=B, Hin=Ho, Win=Wo, Cin=Co, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=1)
    outs = []
    for phase in (1, 0):
        L.lib().tb_gemm_set_variant(9900 + phase)
        dx = torch.zeros(B * H * W, Ci, device="cuda", dtype=torch.float16)
        ops.gemm(nhwc(dy).view(-1, Co), pack_conv_w_dgrad(w), dx, conv=geo, R=R)
        outs.append(dx)
    L.lib().tb_gemm_set_variant(9901)
    ref = nhwc(F.conv_transpose2d(dy.float(), w.float(), stride=2, padding=1, output_padding=1)).reshape(B * H * W, Ci)
    if res:
        ref = ref + R.float()
    parity("phase-ordered transposed dgrad", outs[0], ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
    assert torch.equal(outs[0], outs[1]) or rel_err(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("M,N,K,conv", [(616, 768, 24960, None), (616, 768, 3072, None), (512, 1280, 1280, "8x8"), (2048, 1280, 5120, None)])
def test_split_k_reduced_inside_the_kernel_is_bit_equal_to_the_reducer_launch(M, N, K, conv):
    """tb_gemm_desc.sync (round 3): the k-slice that arrives last at its tile's counter adds the partials in slice order and applies the epilogue --
    the same arithmetic as splitk_reduce_kernel, so the results are bit-equal; the counters are left zeroed; repeated launches reuse them."""
    ops, L = _ops()
    torch.manual_seed(8)
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda")
    if conv:
        B, H, C = 8, 8, 1280
        x = torch.randn(B * H * H, C, device="cuda").half()
        w = (torch.randn(N, 9 * C, device="cuda") / 100).half()
        geo = dict(B=B, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
        run = lambda out: ops.gemm(x, w, out, conv=geo, bias=bias, R=R)  # noqa: E731
    else:
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        run = lambda out: ops.gemm(A, W, out, bias=bias, R=R, act=L.ACT_SILU)  # noqa: E731
    outs = []
    import ctypes
    for knob in (9800, 9801, 9801):
        L.lib().tb_gemm_set_variant(knob)
        out = torch.zeros(M, N, device="cuda")
        run(out)
        cfg = (ctypes.c_int * 5)()
        L.lib().tb_gemm_last_config(cfg)
        outs.append(out)
    L.lib().tb_gemm_set_variant(9800)   # (the default: measured slower in the step)
    torch.cuda.synchronize()
    assert cfg[4] > 1 or L.lib().tb_gemm8_last(None), f"this shape was not split over K (config {list(cfg)})"
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    assert int(ops._gemm_sync_counters(out.device).abs().sum()) == 0


@pytest.mark.parametrize("B,Cin,Cout,H,res,silu", [(8, 1280, 1280, 16, False, True), (8, 2560, 1280, 16, True, True), (8, 1280, 1280, 8, True, False),
                                                   (8, 2560, 1280, 8, False, True), (8, 640, 1280, 16, True, True)])
def test_split_k_slices_added_by_the_groupnorm_behind_the_convolution(B, Cin, Cout, H, res, silu):
    """round 4 (tb_gemm_desc.split_out + tb_groupnorm_fwd_splitk / tb_groupnorm_bwd_splitk): on the 16x16 / 8x8 maps the split-K convolution
    leaves its fp32 slices to the GroupNorm launch behind it, which adds them in slice order, applies the convolution's epilogue (bias, time-
    embedding row bias, residual), writes the convolution's output and normalises -- reducer + GroupNorm arithmetic, so everything is BIT-equal
    to the three-launch sequence; the backward twin never materialises dy."""
    ops, L = _ops()
    torch.manual_seed(21)
    M, HW = B * H * H, H * H
    x = torch.randn(M, Cin, device="cuda").half()
    w = (torch.randn(Cout, 9 * Cin, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    rowbias = torch.randn(B, Cout + 64, device="cuda")[:, 32:32 + Cout]
    R = torch.randn(M, Cout + 8, device="cuda").half()[:, :Cout] if res else None
    gamma, beta = torch.randn(Cout, device="cuda") * 0.5 + 1, torch.randn(Cout, device="cuda") * 0.3
    ws = torch.empty(ops.groupnorm_ws(B, HW, Cout), device="cuda")
    geo = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    assert ops.groupnorm_splitk_ok(B, HW, Cout)
    # three launches: convolution, reducer (inside ops.gemm), GroupNorm
    h_ref = torch.empty(M, Cout, device="cuda", dtype=torch.float16)
    ops.gemm(x, w, h_ref, conv=geo, bias=bias, rowbias=rowbias, rows_per_group=HW, R=R)
    y_ref, st_ref = torch.empty_like(h_ref), torch.empty(B, 32, 2, device="cuda")
    ops.groupnorm_fwd(h_ref, y_ref, gamma, beta, st_ref, ws, B, HW, Cout, silu=silu)
    # two launches
    hbuf = torch.zeros(M, Cout + 16, device="cuda", dtype=torch.float16)
    h = hbuf[:, 8:8 + Cout]
    pk = ops.gemm(x, w, h, conv=geo, bias=bias, rowbias=rowbias, rows_per_group=HW, R=R, defer=True)
    assert isinstance(pk, ops.SplitKPartials) and pk.S > 1, "this shape was not split over K"
    assert float(hbuf.abs().max()) == 0.0   # the convolution's output has NOT been written yet
    y, st = torch.empty_like(h_ref), torch.empty_like(st_ref)
    ops.groupnorm_fwd(h, y, gamma, beta, st, ws, B, HW, Cout, silu=silu, partials=pk)
    assert torch.equal(h, h_ref) and torch.equal(y, y_ref) and torch.equal(st, st_ref)
    assert float(hbuf[:, :8].abs().max()) == 0.0 and float(hbuf[:, 8 + Cout:].abs().max()) == 0.0
    # backward: dgrad convolution (no epilogue) -> GroupNorm backward (+ add)
    dy = torch.randn(M, Cout, device="cuda").half()
    wd = (torch.randn(Cin, 9 * Cout, device="cuda") / (3 * Cout ** 0.5)).half()
    gin, bin_ = torch.randn(Cin, device="cuda") * 0.5 + 1, torch.randn(Cin, device="cuda") * 0.3
    geo_d = dict(B=B, Hin=H, Win=H, Cin=Cout, Hout=H, Wout=H, stride=1, sign=-1, upsample=0, transposed=0)
    if ops.groupnorm_splitk_ok(B, HW, Cin):
        stx, a1 = torch.empty(B, 32, 2, device="cuda"), torch.empty(M, Cin, device="cuda", dtype=torch.float16)
        ops.groupnorm_fwd(x, a1, gin, bin_, stx, ws, B, HW, Cin, silu=silu)
        add = torch.randn(M, Cin, device="cuda").half()
        da_ref = torch.empty(M, Cin, device="cuda", dtype=torch.float16)
        ops.gemm(dy, wd, da_ref, conv=geo_d)
        dx_ref = torch.empty_like(da_ref)
        ops.groupnorm_bwd(da_ref, x, gin, bin_, stx, dx_ref, ws, B, HW, Cin, silu=silu, add=add)
        da = torch.zeros_like(da_ref)
        pk = ops.gemm(dy, wd, da, conv=geo_d, defer=True)
        if isinstance(pk, ops.SplitKPartials):   # (1280 -> 2560 at 16x16 has enough tiles and is not split)
            dx = torch.empty_like(dx_ref)
            ops.groupnorm_bwd(da, x, gin, bin_, stx, dx, ws, B, HW, Cin, silu=silu, add=add, partials=pk)
            assert torch.equal(dx, dx_ref) and float(da.abs().max()) == 0.0
        else:
            assert pk is da and torch.equal(da, da_ref) and Cin == 2560 and H == 16
    # a launch that is NOT split writes its output as usual and returns it
    xs = torch.randn(8 * 64 * 64, 320, device="cuda").half()
    ws_ = (torch.randn(320, 9 * 320, device="cuda") / 50).half()
    o = torch.empty(8 * 64 * 64, 320, device="cuda", dtype=torch.float16)
    r = ops.gemm(xs, ws_, o, conv=dict(B=8, Hin=64, Win=64, Cin=320, Hout=64, Wout=64, stride=1, sign=1, upsample=0, transposed=0), defer=True)
    assert r is o


def test_unet_with_and_without_deferred_split_k_reduction_is_bit_equal():
    """the whole executor (SD-shaped tiny UNet at B = 8, where the 8x8 / 4x4 maps split K): forward and dgrad backward with the reducer launches
    (TB_DEFER_SPLITK=0 behaviour) and with the GroupNorm launches adding the slices must give identical bits"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_model import make_unet
    ops, L = _ops()
    B, hw, D = 8, 32, 64
    ref, hip, cfg = make_unet(B, hw, D, channels=(64, 128, 256, 256))   # 8-channel groups on the 8x8 / 4x4 maps: the one-pass GroupNorm kernels
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, hw, hw, generator=g).half().cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    ehs = torch.randn(B * 77, D, generator=g).half().cuda()
    dpred = torch.randn(B, 4, hw, hw, generator=g).cuda()
    outs = []
    prev = ops.DEFER_SPLITK
    try:
        for flag in (False, True):
            ops.DEFER_SPLITK = flag
            ops.start_recording()
            pred = hip.forward(x, t, ehs).clone()
            d_ehs = hip.backward(dpred).clone()
            names = [r[0] for r in ops.stop_recording()]
            outs.append((pred, d_ehs, names))
    finally:
        ops.DEFER_SPLITK = prev
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not any("splitk" in n for n in outs[0][2])
    nf, nb = sum(n == "groupnorm_fwd(splitk)" for n in outs[1][2]), sum(n == "groupnorm_bwd(splitk)" for n in outs[1][2])
    print(f"[deferred split-K] {nf} forward and {nb} backward GroupNorm launches add their producer's slices")
    assert nf >= 4 and nb >= 4


@pytest.mark.parametrize("B,C,Hc", [(8, 640, 32), (8, 1280, 16), (2, 640, 32), (4, 320, 64), (8, 1280, 32)])
def test_upsampler_convolution_as_sub_pixel_convolutions(B, C, Hc):
    """round 4 (csrc/gemm8.hip SUB modes; diffusers Upsample2D = nearest x2 + conv3x3): the forward as four 2x2-tap convolutions on the coarse map
    (tb_gemm_desc.upsample = 2) and its input gradient from the fine gradient's four strided views (upsample = 3), against torch on the upsampled
    map and against the 9-tap kernels on the materialised 4x map.  Stated divergence: the pre-summed filter rows are rounded to fp16 once
    (relative 2^-11 per weight) -- bounded here by the same 2e-3 / 4e-3 / 3e-3 triple as every other convolution."""
    ops, L = _ops()
    torch.manual_seed(31)
    Hf = 2 * Hc
    if not ops.subpixel_ok(B, Hc, Hc, C, C):
        pytest.skip("shape not covered by the sub-pixel tiles")
    x = torch.randn(B, C, Hc, Hc, device="cuda").half()
    w = (torch.randn(C, C, 3, 3, device="cuda") / (3 * C ** 0.5)).half()
    bias = torch.randn(C, device="cuda")
    wf, wd = ops.pack_subpixel_weights(w)
    xn = nhwc(x).reshape(B * Hc * Hc, C)
    out = torch.empty(B * Hf * Hf, C, device="cuda", dtype=torch.float16)
    ops.gemm(xn, wf, out, bias=bias, conv=dict(B=B, Hin=Hc, Win=Hc, Cin=C, Hout=Hf, Wout=Hf, stride=1, sign=1, upsample=2, transposed=0))
    xr = x.float().requires_grad_(True)
    ref = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), w.float(), bias, padding=1)
    parity(f"sub-pixel upsampler forward {B}x{C}x{Hc}", out.view(B, Hf, Hf, C), nhwc(ref), rel=2e-3, maxabs=4e-3, ch_dim=3, ch_rel=3e-3)
    # the 9-tap path on the materialised map
    xu = torch.empty(B * Hf * Hf, C, device="cuda", dtype=torch.float16)
    ops.upsample2x(xn, xu, B, Hc, Hc, C)
    out9 = torch.empty_like(out)
    ops.gemm(xu, pack_conv_w(w), out9, bias=bias, conv=dict(B=B, Hin=Hf, Win=Hf, Cin=C, Hout=Hf, Wout=Hf, stride=1, sign=1, upsample=0, transposed=0))
    assert rel_err(out, out9) < 1.5e-3
    # input gradient
    dy = torch.randn(B, C, Hf, Hf, device="cuda").half()
    dyn = nhwc(dy).reshape(B * Hf * Hf, C)
    dx = torch.empty(B * Hc * Hc, C, device="cuda", dtype=torch.float16)
    ops.gemm(dyn, wd, dx, conv=dict(B=B, Hin=Hf, Win=Hf, Cin=C, Hout=Hc, Wout=Hc, stride=1, sign=1, upsample=3, transposed=0))
    ref.backward(dy.float())
    parity(f"sub-pixel upsampler dgrad {B}x{C}x{Hc}", dx.view(B, Hc, Hc, C), nhwc(xr.grad), rel=2e-3, maxabs=4e-3, ch_dim=3, ch_rel=3e-3)
    du = torch.empty(B * Hf * Hf, C, device="cuda", dtype=torch.float16)
    ops.gemm(dyn, pack_conv_w_dgrad(w), du, conv=dict(B=B, Hin=Hf, Win=Hf, Cin=C, Hout=Hf, Wout=Hf, stride=1, sign=-1, upsample=0, transposed=0))
    dx9 = torch.empty_like(dx)
    ops.pool2x2_sum(du, dx9, B, Hc, Hc, C)
    assert rel_err(dx, dx9) < 2e-3   # (the 9-tap path rounds the fine-map gradient to fp16 before pooling)
    # shapes the tiles do not cover are refused, not silently mis-computed
    bad = torch.empty(2 * 16 * 16, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(2 * 8 * 8, 64, device="cuda", dtype=torch.float16), torch.zeros(256, 256, device="cuda", dtype=torch.float16), bad,
                 conv=dict(B=2, Hin=8, Win=8, Cin=64, Hout=16, Wout=16, stride=1, sign=1, upsample=2, transposed=0))


def test_sub_pixel_descs_never_reach_the_four_wave_kernel():
    """with the 8-wave family switched off (tb_gemm8_set(0), the documented A/B switch) a sub-pixel desc must be refused -- the 4-wave
    kernel would read `upsample != 0` as the folded 9-tap gather over a K = 4 Cin / 16 Cin weight -- and `subpixel_ok` must say so, so that HipUNet
    keeps the 9-tap form."""
    ops, L = _ops()
    B, C, Hc = 8, 640, 32
    assert ops.subpixel_ok(B, Hc, Hc, C, C)
    w = (torch.randn(C, C, 3, 3, device="cuda") / (3 * C ** 0.5)).half()
    wf, wd = ops.pack_subpixel_weights(w)
    xn = torch.randn(B * Hc * Hc, C, device="cuda").half()
    out = torch.empty(B * 4 * Hc * Hc, C, device="cuda", dtype=torch.float16)
    dx = torch.empty(B * Hc * Hc, C, device="cuda", dtype=torch.float16)
    old = L.lib().tb_gemm8_set(0)
    try:
        assert not ops.subpixel_ok(B, Hc, Hc, C, C)
        with pytest.raises(RuntimeError):
            ops.gemm(xn, wf, out, conv=dict(B=B, Hin=Hc, Win=Hc, Cin=C, Hout=2 * Hc, Wout=2 * Hc, stride=1, sign=1, upsample=2, transposed=0))
        with pytest.raises(RuntimeError):
            ops.gemm(out, wd, dx, conv=dict(B=B, Hin=2 * Hc, Win=2 * Hc, Cin=C, Hout=Hc, Wout=Hc, stride=1, sign=1, upsample=3, transposed=0))
    finally:
        L.lib().tb_gemm8_set(old)
    assert ops.subpixel_ok(B, Hc, Hc, C, C)


@pytest.mark.parametrize("M,N,bias,res", [(32768, 320, True, True), (32768, 960, False, False), (25600, 1280, True, False), (32768, 128, False, True)])
def test_linear_k320_activation_stationary_kernel(M, N, bias, res):
    """csrc/lin320.hip (round 4): the K = 320 Linear layers of the 64x64 maps with the 128-row activation tile register-resident, against torch
    and against the wide-tile kernel it replaces (knob 9400); strided output / residual views."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(13)
    K = 320
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda") if bias else None
    Rbuf = torch.randn(M, N + 8, device="cuda").half() if res else None
    R = Rbuf[:, 4:4 + N] if res else None
    outs = []
    for knob in (9401, 9400):
        L.lib().tb_gemm_set_variant(knob)
        Cbuf = torch.full((M, N + 16), 3.0, device="cuda", dtype=torch.float16)
        out = Cbuf[:, 8:8 + N]
        ops.gemm(A, W, out, bias=b, R=R)
        cfg = (ctypes.c_int * 5)()
        L.lib().tb_gemm_last_config(cfg)
        took_lin320 = cfg[3] == 643 and not L.lib().tb_gemm8_last(None)      # {128, 64, 3, 643, 1} names the lin320 launch (an 8-wave launch leaves it stale)
        assert took_lin320 == (knob == 9401), list(cfg)
        assert (Cbuf[:, :8] == 3).all() and (Cbuf[:, 8 + N:] == 3).all()
        outs.append(out)
    L.lib().tb_gemm_set_variant(9401)
    ref = A.float() @ W.float().T + (b if bias else 0) + (R.float() if res else 0)
    parity("K = 320 activation-stationary Linear", outs[0], ref, rel=1e-3, maxabs=2e-3, ch_dim=1, ch_rel=2e-3)
    assert torch.equal(outs[0], outs[1]) or rel_err(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("M,N,K,bias,res,act", [(2048, 1280, 1280, True, True, 0), (2048, 1280, 640, False, False, 0), (2048, 1280, 2560, True, False, 0),
                                                (2048, 1280, 1344, True, True, 1), (2048, 1280, 3840, False, False, 0)])   # (K = 3840: the qkv dgrad)
def test_linear_one_tile_per_cu_128x80(M, N, K, bias, res, act):
    """the 16x16-map Linear layers as ONE 128 x 80 tile per CU on a 4-stage ring (two stages in flight across the barrier): round 5's
    gemm8_kernel<4, 1, 2, 5, false, 4, 0, 2> (4 x 1 x 2 waves: the two waves of a SIMD split every k-step, accumulators added through the LDS),
    against torch, against round 4's 8 x 1 waves (tb_gemm8_set bit 16384) and against the 4-wave kernels that took these shapes before (bit 2048);
    strided views."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(17)
    A = torch.randn(M, K + 8, device="cuda").half()[:, :K]
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda") if bias else None
    R = torch.randn(M, N + 8, device="cuda").half()[:, 8:] if res else None
    prev = L.lib().tb_gemm8_set(39)
    outs = []
    try:
        for bits, tile in ((39, [4, 1, 2, 5]), (39 | 16384, [8, 1, 1, 5]), (39 | 2048, None)):
            L.lib().tb_gemm8_set(bits)
            Cbuf = torch.full((M, N + 16), 3.0, device="cuda", dtype=torch.float16)
            out = Cbuf[:, 8:8 + N]
            ops.gemm(A, W, out, bias=b, R=R, act=L.ACT_SILU if act else L.ACT_NONE)
            last = (ctypes.c_int * 6)()
            took = bool(L.lib().tb_gemm8_last(last))
            assert (took and list(last)[:4] == tile) if tile else not (took and list(last)[:4] in ([4, 1, 2, 5], [8, 1, 1, 5])), list(last)
            assert (Cbuf[:, :8] == 3).all() and (Cbuf[:, 8 + N:] == 3).all()
            outs.append(out)
    finally:
        L.lib().tb_gemm8_set(prev)
    ref = A.float() @ W.float().T + (b if bias else 0) + (R.float() if res else 0)
    if act:
        ref = F.silu(ref)      # (tb_gemm: the activation acts on acc + bias + residual)
    parity("128 x 80 one-per-CU Linear tile, k-halves", outs[0], ref, rel=1e-3, maxabs=4e-3, ch_dim=1, ch_rel=2e-3)
    parity("128 x 80 one-per-CU Linear tile, 8 x 1 waves", outs[1], ref, rel=1e-3, maxabs=4e-3, ch_dim=1, ch_rel=2e-3)
    assert rel_err(outs[0], outs[1]) < 3e-4 and rel_err(outs[0], outs[2]) < 3e-4


@pytest.mark.parametrize("B,C,Hc,res", [(8, 320, 32, True), (8, 640, 16, True), (8, 320, 32, False), (2, 320, 32, True)])
def test_stride2_conv_dgrad_as_sub_pixel_convolution(B, C, Hc, res):
    """round 5: the input gradient of `Downsample2D`'s stride-2 convolution written as a sub-pixel convolution of d out
    (`ops.pack_strided_dgrad_subpixel`, `conv=dict(upsample=2)`, gemm8 SUB = 1): 9 of its 16 (class, tap) blocks carry one filter tap, 7 are zero --
    the convolution's own weights, nothing summed.  Against conv_transpose2d and against the 4-wave transposed gather it replaces, with the skip
    gradient riding in as the residual."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(29)
    H = 2 * Hc
    w = (torch.randn(C, C, 3, 3, device="cuda") / (3 * C ** 0.5)).half()
    dy = torch.randn(B, C, Hc, Hc, device="cuda").half()
    R = torch.randn(B * H * H, C + 8, device="cuda").half()[:, 8:] if res else None
    if not ops.subpixel_ok(B, Hc, Hc, C, C):
        pytest.skip("coarse map does not tile")
    dx = torch.full((B * H * H, C + 16), 3.0, device="cuda", dtype=torch.float16)
    out = dx[:, 8:8 + C]
    ops.gemm(nhwc(dy).view(-1, C), ops.pack_strided_dgrad_subpixel(w), out, R=R,
             conv=dict(B=B, Hin=Hc, Win=Hc, Cin=C, Hout=H, Wout=H, stride=1, sign=1, upsample=2, transposed=0))
    last = (ctypes.c_int * 6)()
    assert L.lib().tb_gemm8_last(last) and list(last)[4] == 1, list(last)
    assert (dx[:, :8] == 3).all() and (dx[:, 8 + C:] == 3).all()
    old = torch.zeros(B * H * H, C, device="cuda", dtype=torch.float16)
    ops.gemm(nhwc(dy).view(-1, C), pack_conv_w_dgrad(w), old, R=R,
             conv=dict(B=B, Hin=Hc, Win=Hc, Cin=C, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=1))
    ref = nhwc(F.conv_transpose2d(dy.float(), w.float(), stride=2, padding=1, output_padding=1)).reshape(B * H * H, C)
    if res:
        ref = ref + R.float()
    parity("stride-2 dgrad as sub-pixel convolution", out, ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
    assert rel_err(out, old) < 3e-4, rel_err(out, old)


@pytest.mark.parametrize("act", ["quick_gelu", "quick_gelu_grad", "none_f32"])
def test_linear_ragged_last_row_tile_128x128(act):
    """the text encoder's wide layers (M = 24 x 77 = 1848 token rows: not a multiple of any tile height; N = 3072, K = 768) on the 8-wave 128 x 128
    tile with a RAGGED last row tile (tb_gemm8_set bit 131072, an experiment: 28.3 -> 27.4 us, not worth a default): the panel loads of rows past M
    read row 0 and nothing of them is stored (canary rows), the generic epilogue (GELU variants with the saved pre-activation, fp32 output +
    fp32 residual) agrees with the 4-wave kernel and with torch."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(23)
    M, N, K = 1848, 3072, 768
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    pre = torch.randn(M, N, device="cuda").half()
    R32 = torch.randn(M, N, device="cuda")
    prev = L.lib().tb_gemm8_set(39)
    outs, pres = [], []
    try:
        for bits in (39 | 131072, 39):
            L.lib().tb_gemm8_set(bits)
            f32 = act == "none_f32"
            Cbuf = torch.full((M + 128, N), 3.0, device="cuda", dtype=torch.float32 if f32 else torch.float16)
            out = Cbuf[:M]
            c2 = pre.clone() if act == "quick_gelu_grad" else (torch.zeros_like(pre) if act == "quick_gelu" else None)
            if act == "quick_gelu":
                ops.gemm(A, W, out, bias=b, act=L.ACT_QUICK_GELU, C2=c2)
            elif act == "quick_gelu_grad":
                ops.gemm(A, W, out, act=L.ACT_QUICK_GELU_GRAD, C2=c2)
            else:
                ops.gemm(A, W, out, bias=b, R=R32)
            last = (ctypes.c_int * 6)()
            took = bool(L.lib().tb_gemm8_last(last))
            assert (took and list(last)[:4] == [2, 4, 4, 2]) == (bits != 39), list(last)
            assert (Cbuf[M:] == 3).all()
            outs.append(out)
            pres.append(c2)
    finally:
        L.lib().tb_gemm8_set(prev)
    acc = A.float() @ W.float().T
    if act == "quick_gelu":
        z = acc + b
        ref = z * torch.sigmoid(1.702 * z)
        assert rel_err(pres[0], z) < 1e-3 and torch.equal(pres[0], pres[1])
    elif act == "quick_gelu_grad":
        z = pre.float()
        sg = torch.sigmoid(1.702 * z)
        ref = acc * (sg * (1 + 1.702 * z * (1 - sg)))
    else:
        ref = acc + b + R32
    parity("ragged 128 x 128 Linear tile", outs[0], ref, rel=1e-3, maxabs=6e-3, ch_dim=1, ch_rel=2e-3)
    assert rel_err(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("K,bias,res", [(640, True, True), (2560, True, True), (3840, False, False), (5120, False, True)])
def test_linear_one_tile_per_cu_128x160(K, bias, res):
    """the 32x32-map Linear layers (M = 8192, N = 640: 64 x 4 tiles of 128 x 160, one per CU, 4-stage ring) up to K = 5120 (round 5: the
    GEGLU-projection dgrad, K = 8 C, used to fall onto 64 x 320 tiles of the 3-stage ring -- 85 -> 66 us): against torch and against the route the
    shape took before (tb_gemm8_set bit 262144 = the old K <= 2560 limit; bit 4096 = no 128 x 160 one-per-CU tile at all)."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(31)
    M, N = 8192, 640
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda") if bias else None
    R = torch.randn(M, N + 8, device="cuda").half()[:, 8:] if res else None
    prev = L.lib().tb_gemm8_set(39)
    outs = []
    try:
        for bits in (39, 39 | 262144 | 4096):
            L.lib().tb_gemm8_set(bits)
            Cbuf = torch.full((M, N + 16), 3.0, device="cuda", dtype=torch.float16)
            out = Cbuf[:, 8:8 + N]
            ops.gemm(A, W, out, bias=b, R=R)
            last = (ctypes.c_int * 6)()
            took = bool(L.lib().tb_gemm8_last(last))
            if bits == 39:
                assert took and list(last)[:4] == [4, 2, 2, 5] and list(last)[5] == 4, list(last)
            assert (Cbuf[:, :8] == 3).all() and (Cbuf[:, 8 + N:] == 3).all()
            outs.append(out)
    finally:
        L.lib().tb_gemm8_set(prev)
    ref = A.float() @ W.float().T + (b if bias else 0) + (R.float() if res else 0)
    parity("128 x 160 one-per-CU Linear tile", outs[0], ref, rel=1e-3, maxabs=4e-3, ch_dim=1, ch_rel=2e-3)
    assert rel_err(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("K,bias,res", [(5120, True, True), (10240, False, False)])
def test_linear_long_k_two_slices_of_128x160_tiles(K, bias, res):
    """the 16x16-map long-K Linear layers (M = 2048, N = 1280: ff.net.2, the GEGLU-projection dgrad) as 16 x 8 tiles of 128 x 160 in two k-slices on the
    8-wave kernel (one workgroup per CU, fp32 partials + the reducer launch), against torch and against the 128 x 128 split-K launches of the
    4-wave kernel they replace (tb_gemm8_set bit 8192)."""
    ops, L = _ops()
    import ctypes
    torch.manual_seed(19)
    M, N = 2048, 1280
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda") if bias else None
    R = torch.randn(M, N + 8, device="cuda").half()[:, 8:] if res else None
    prev = L.lib().tb_gemm8_set(39)
    outs = []
    try:
        for bits in (39, 39 | 8192):
            L.lib().tb_gemm8_set(bits)
            Cbuf = torch.full((M, N + 16), 3.0, device="cuda", dtype=torch.float16)
            out = Cbuf[:, 8:8 + N]
            ops.gemm(A, W, out, bias=b, R=R)
            last = (ctypes.c_int * 6)()
            took = bool(L.lib().tb_gemm8_last(last))
            assert (took and list(last)[:4] == [4, 2, 2, 5]) == (bits == 39), list(last)
            assert (Cbuf[:, :8] == 3).all() and (Cbuf[:, 8 + N:] == 3).all()
            outs.append(out)
    finally:
        L.lib().tb_gemm8_set(prev)
    ref = A.float() @ W.float().T + (b if bias else 0) + (R.float() if res else 0)
    parity("long-K Linear, two k-slices of 128 x 160 tiles", outs[0], ref, rel=1e-3, maxabs=4e-3, ch_dim=1, ch_rel=2e-3)
    assert rel_err(outs[0], outs[1]) < 3e-4


@pytest.mark.parametrize("M,C", [(8192, 640), (2048, 1280)])
def test_layernorm_folded_into_the_consuming_linear(M, C):
    """round 5 (tb_gemm_desc.rs_out / rs_in; torch.nn.LayerNorm norm1 / norm2 / norm3 of diffusers BasicTransformerBlock in front of attn1.to_q/k/v,
    attn2.to_q, ff.net.0.proj -- train_textboost.py:1063-1067): the producer of the residual stream writes per-column-tile (sum, sum of squares) of
    its fp16 output rows; the Linear behind the LayerNorm multiplies the RAW stream by gamma-folded weights and applies (mean, rstd) in its epilogue.
    Against torch fp32 (LayerNorm -> Linear / GEGLU) and against the launches it replaces (tb_layernorm_fwd + tb_gemm); non-zero row means and a
    gamma / beta far from (1, 0) so that a dropped mean, c1 or c2 term fails."""
    ops, L = _ops()
    torch.manual_seed(41)
    slots = ops.lnfold_slots(M, C)
    assert slots > 0
    o = torch.randn(M, C, device="cuda").half()
    Wo = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
    bo = torch.randn(C, device="cuda")
    R = (torch.randn(M, C, device="cuda") * 0.7 + torch.randn(M, 1, device="cuda")).half()     # a row mean of order 1
    gamma, beta = 1 + 0.5 * torch.randn(C, device="cuda"), 0.3 * torch.randn(C, device="cuda")
    # ---- producer: x = o Wo^T + bo + R, plus the row statistics
    x = torch.empty(M, C, device="cuda", dtype=torch.float16)
    rs = torch.full((M, 16, 2), 7.0, device="cuda")
    ops.gemm(o, Wo, x, bias=bo, R=R, rs_out=rs)
    x_plain = torch.empty_like(x)
    ops.gemm(o, Wo, x_plain, bias=bo, R=R)
    assert torch.equal(x, x_plain)                                   # the statistics do not change the product
    assert (rs[:, slots:] == 7.0).all()                              # only the slots in use are written
    tw = C // slots
    xs = x.float().view(M, slots, tw)
    assert torch.allclose(rs[:, :slots, 0], xs.sum(-1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(rs[:, :slots, 1], (xs * xs).sum(-1), rtol=1e-5, atol=1e-3)
    ln_ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    mean_ref, rstd_ref = x.float().mean(1), (x.float().var(1, unbiased=False) + 1e-5).rsqrt()
    # the launches the fold replaces: tb_layernorm_fwd -> tb_gemm
    l = torch.empty_like(x)
    st_old = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, l, gamma, beta, st_old)
    # ---- consumers: qkv (N = 3C, no bias), attn2.to_q (N = C), the GEGLU projection (N = 8C, packed rows)
    for N, has_bias in ((3 * C, False), (C, False)):
        W = (torch.randn(N, C, device="cuda") / C ** 0.5).half()
        b = torch.randn(N, device="cuda") if has_bias else None
        Wp, c1, c2 = ops.fold_layernorm(W, gamma, beta, b)
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        st = torch.zeros(M, 2, device="cuda")
        ops.gemm(x, Wp, out, bias=c2, lnfold=(rs, slots, c1, st, 1e-5))
        ref = ln_ref @ W.float().T + (b if b is not None else 0)
        parity(f"folded LN -> Linear {M}x{N}x{C}", out, ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
        old = torch.empty_like(out)
        ops.gemm(l, W, old, bias=b)
        assert rel_err(out, old) < 2e-3
        assert torch.allclose(st[:, 0], mean_ref, rtol=1e-4, atol=1e-4) and torch.allclose(st[:, 1], rstd_ref, rtol=2e-4, atol=1e-5)
        assert torch.allclose(st, st_old, rtol=3e-4, atol=2e-4)
    from textboost_amd.unet import pack_geglu_rows
    Wff = torch.randn(8 * C, C, device="cuda") / C ** 0.5
    bff = torch.randn(8 * C, device="cuda")
    Wp, c1, c2 = ops.fold_layernorm(pack_geglu_rows(Wff), gamma, beta, pack_geglu_rows(bff))
    gated = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    raw = torch.empty(M, 8 * C, device="cuda", dtype=torch.float16)
    st = torch.zeros(M, 2, device="cuda")
    ops.gemm(x, Wp, gated, bias=c2, act=L.ACT_GEGLU, C2=raw, lnfold=(rs, slots, c1, st, 1e-5))
    proj = ln_ref @ Wff.half().float().T + bff
    h, g = proj.chunk(2, dim=1)
    parity(f"folded LN -> GEGLU {M}x{8 * C}x{C}", gated, h.half().float() * F.gelu(g.half().float()), rel=3e-3, maxabs=6e-3, ch_dim=1, ch_rel=4e-3)
    parity("folded LN -> GEGLU pre-gate projections", raw, pack_geglu_rows(proj.T.contiguous()).T, rel=2e-3, maxabs=4e-3)
    gated_old, raw_old = torch.empty_like(gated), torch.empty_like(raw)
    ops.gemm(l, pack_geglu_rows(Wff).half(), gated_old, bias=pack_geglu_rows(bff), act=L.ACT_GEGLU, C2=raw_old)
    assert rel_err(gated, gated_old) < 3e-3 and rel_err(raw, raw_old) < 2e-3
    assert torch.allclose(st[:, 1], rstd_ref, rtol=2e-4, atol=1e-5)
    # backward: the dgrad through W' carries gamma, so tb_layernorm_bwd runs with gamma = 1
    dy = torch.randn(M, C, device="cuda").half()
    W = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
    Wp, c1, c2 = ops.fold_layernorm(W, gamma, beta)
    g_f = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.gemm(dy, Wp.t().contiguous(), g_f)
    dx = torch.empty(M, C, device="cuda", dtype=torch.float16)
    ops.layernorm_bwd(g_f, x, torch.ones(C, device="cuda"), st, dx)
    xr = x.float().requires_grad_(True)
    (F.layer_norm(xr, (C,), gamma, beta, 1e-5) @ W.float().T).backward(dy.float())
    parity("folded LN backward", dx, xr.grad, rel=3e-3, maxabs=6e-3)
    # requests the tiles cannot serve are refused
    with pytest.raises(RuntimeError):
        ops.gemm(o[:512], Wo, x[:512], bias=bo, R=R[:512], rs_out=rs[:512].contiguous())      # 4 row tiles: not an 8-wave one-per-CU launch


@pytest.mark.parametrize("M,C", [(8192, 640), (2048, 1280)])
def test_layernorm_fold_with_a_large_row_mean_and_its_refusals(M, C):
    """ The folded LayerNorm takes the row variance as E[x^2] - mean^2 in fp32 from the producer's tile sums and multiplies the RAW
    stream: a residual stream whose row mean is ~50 standard deviations (the regime where a one-pass variance loses digits) must still match
    torch's LayerNorm -> Linear.  And the protocol's refusals: a producer whose column tiles would fill another slot count than its consumer sums,
    or more slots than a statistics row holds (tb_gemm8.hip launch8); a consumer while a profiling knob has its lean epilogue off (gemm.hip)."""
    ops, L = _ops()
    torch.manual_seed(43)
    slots = ops.lnfold_slots(M, C)
    o = torch.randn(M, C, device="cuda").half()
    Wo = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
    R = (50.0 + 0.1 * torch.randn(M, 1, device="cuda")).expand(M, C).half()          # row mean ~50, row std ~1
    gamma, beta = 1 + 0.5 * torch.randn(C, device="cuda"), 0.3 * torch.randn(C, device="cuda")
    x = torch.empty(M, C, device="cuda", dtype=torch.float16)
    rs = torch.zeros(M, 16, 2, device="cuda")
    ops.gemm(o, Wo, x, R=R, rs_out=rs, rs_slots=slots)
    xf = x.float()
    assert 30 < (xf.mean(1).abs() / xf.std(1)).median().item() < 80
    ln_ref = F.layer_norm(xf.double(), (C,), gamma.double(), beta.double(), 1e-5).float()
    W = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).half()
    Wp, c1, c2 = ops.fold_layernorm(W, gamma, beta, None)
    out = torch.empty(M, 3 * C, device="cuda", dtype=torch.float16)
    st = torch.zeros(M, 2, device="cuda")
    ops.gemm(x, Wp, out, bias=c2, lnfold=(rs, slots, c1, st, 1e-5))
    parity(f"folded LN -> Linear, row mean / std ~ 50, {M}x{3 * C}x{C}", out, ln_ref @ W.float().T, rel=3e-3, maxabs=6e-3, ch_dim=1, ch_rel=4e-3)
    rstd_ref = (xf.double().var(1, unbiased=False) + 1e-5).rsqrt().float()
    assert torch.allclose(st[:, 1], rstd_ref, rtol=2e-3, atol=0) and torch.allclose(st[:, 0], xf.mean(1), rtol=1e-5, atol=1e-3)
    # ---- refusals
    with pytest.raises(RuntimeError):
        ops.gemm(o, Wo, x, R=R, rs_out=rs, rs_slots=slots + 1)                      # the consumer would sum another number of slots
    with pytest.raises(RuntimeError):
        ops.gemm(o, Wo, x, R=R, rs_out=torch.zeros(M, slots - 1, 2, device="cuda"))  # a statistics row narrower than the tile grid
    L.lib().tb_gemm_set_variant(2000 + 8)                                            # profiling knob: the 4-wave kernels' lean epilogue off
    try:
        if C == 1280:   # (the 16x16-map qkv projection is the consumer that runs on the 4-wave tiles)
            with pytest.raises(RuntimeError):
                ops.gemm(x, Wp, out, bias=c2, lnfold=(rs, slots, c1, st, 1e-5))
    finally:
        L.lib().tb_gemm_set_variant(2000)                                            # (2000 + bits: all profiling bits off again)


def test_conv3x3_32x32_maps_as_two_k_slices_of_the_wide_tile():
    """round-6 experiment behind tb_gemm8_set bit 524288 (measured a LOSS in the step, 28.12 -> 28.27 ms, so off by default): the 32x32-map
    convolutions (M = 8192, N = 640: 128 tiles of 256 x 160) as two k-slices of the 256 x 160 tile + the reducer, instead of one round of
    256 x 80 tiles.  Same result as the default path and as F.conv2d."""
    import ctypes
    ops, L = _ops()
    torch.manual_seed(12)
    B, Cin, Cout, H = 8, 640, 640, 32
    x = torch.randn(B, Cin, H, H, device="cuda").half()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(B * H * H, Cout, device="cuda").half()
    xn = nhwc(x).view(B * H * H, Cin)
    geo = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    default_bits = L.lib().tb_gemm8_set(39)
    outs = []
    try:
        for bits, want in ((39, [8, 1, 2, 5, 1, 4]), (39 | 524288, [4, 2, 4, 5, 1, 3])):
            L.lib().tb_gemm8_set(bits)
            out = torch.empty(B * H * H, Cout, device="cuda", dtype=torch.float16)
            ops.gemm(xn, pack_conv_w(w), out, bias=bias, R=res, conv=geo)
            last = (ctypes.c_int * 6)()
            assert L.lib().tb_gemm8_last(last) and list(last) == want, list(last)
            outs.append(out)
    finally:
        L.lib().tb_gemm8_set(default_bits)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * H, Cout) + res.float()
    parity("32x32-map conv as two k-slices of the 256 x 160 tile", outs[1], ref, rel=2e-3, maxabs=4e-3, ch_dim=1, ch_rel=3e-3)
    assert rel_err(outs[0], outs[1]) < 1e-3


def test_conv3x3_wide_tile_as_one_stream_per_wave_equals_the_phase_form():
    """round 6, tb_gemm8_set bit 1048576 (opt-in: isolated launches -4 ... -6 %, the sustained step 27.92 -> 27.99 ms -- these tiles sit at the board's
    power limit, see launch8): the 256 x 160 and 256 x 128 convolution tiles with the next k-step's fragment reads between this step's MFMAs and
    one barrier per tap.  Both forms add the same products in the same order: bit-equal outputs -- forward with bias + residual, dgrad (flipped
    taps), the split-K slices of the 16x16 maps, the 128-wide tile -- and both match torch."""
    import ctypes
    ops, L = _ops()
    torch.manual_seed(13)
    cases = [(8, 320, 320, 64, 1, [4, 2, 4, 5, 1, 3]), (8, 320, 320, 64, -1, [4, 2, 4, 5, 1, 3]), (2, 960, 320, 64, 1, [4, 2, 4, 5, 1, 3]),
             (8, 1280, 1280, 16, 1, [4, 2, 4, 5, 1, 3]), (8, 128, 256, 64, 1, None)]
    default_bits = L.lib().tb_gemm8_set(39)
    try:
        for B, Cin, Cout, H, sign, want in cases:
            w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).half()
            if sign > 0:
                x = torch.randn(B, Cin, H, H, device="cuda").half()
                bias = torch.randn(Cout, device="cuda")
                res = torch.randn(B * H * H, Cout, device="cuda").half()
                a, wp, n_in, n_out = nhwc(x).view(B * H * H, Cin), pack_conv_w(w), Cin, Cout
                ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * H, Cout) + res.float()
            else:
                dy = torch.randn(B, Cout, H, H, device="cuda").half()
                bias, res = None, None
                a, wp, n_in, n_out = nhwc(dy).view(B * H * H, Cout), pack_conv_w_dgrad(w), Cout, Cin
                ref = nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1)).view(B * H * H, Cin)
            geo = dict(B=B, Hin=H, Win=H, Cin=n_in, Hout=H, Wout=H, stride=1, sign=sign, upsample=0, transposed=0)
            outs = []
            for bits in (39, 39 | 1048576):
                L.lib().tb_gemm8_set(bits)
                out = torch.empty(B * H * H, n_out, device="cuda", dtype=torch.float16)
                ops.gemm(a, wp, out, bias=bias, R=res, conv=geo)
                last = (ctypes.c_int * 6)()
                used = L.lib().tb_gemm8_last(last)
                if want is not None:
                    assert used and list(last) == want, (B, Cin, Cout, H, list(last))
                outs.append(out)
            parity(f"stream conv {Cin}->{Cout}@{H} sign {sign}", outs[1], ref, rel=2e-3, maxabs=6e-3, ch_dim=1, ch_rel=3e-3)
            assert torch.equal(outs[0], outs[1]), (B, Cin, Cout, H, sign, rel_err(outs[0], outs[1]))
    finally:
        L.lib().tb_gemm8_set(default_bits)
