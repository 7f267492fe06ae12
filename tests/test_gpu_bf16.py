"""GPU parity of the bfloat16 build of the kernel library (libtextboost_hip_bf16.so: the same sources with -DTB_BF16) -- the reference's
`--mixed_precision bf16` (train_textboost.py:298-308 choices no|fp16|bf16, :930-934 weight_dtype = bf16; accelerate creates a GradScaler for
fp16 only).  Kernels against torch fp32 on the bf16-rounded operands, the executors and a whole optimizer step against the fp32 oracle.

Tolerances: bf16 carries 8 significand bits against fp16's 11, so every bound of the fp16 tests is relaxed by 8x (rounding noise scales with
2^-8 / 2^-11); accumulation and statistics stay fp32 as in the fp16 build."""
import os

import pytest
import torch
import torch.nn.functional as F

from parity import parity

pytestmark = pytest.mark.gpu
dev = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def bf16_library():
    from textboost_amd import _lib as L
    prev = L.set_half("bf16")
    assert L.half_dtype() == BF
    yield
    L.set_half(prev)


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def test_bf16_library_is_a_second_build_with_the_same_abi():
    from textboost_amd import _lib as L
    assert os.path.basename(L.LIB_PATH_BF16).startswith("libtextboost_hip_bf16") and os.path.exists(L.LIB_PATH_BF16)
    h = L.lib()
    assert L.half_kind() == "bf16" and h is not None
    with pytest.raises(AssertionError):      # an fp16 tensor handed to the bf16 library is refused at the boundary
        from textboost_amd import ops
        ops.gemm(torch.zeros(64, 64, device=dev, dtype=torch.float16), torch.zeros(64, 64, device=dev, dtype=torch.float16),
                 torch.zeros(64, 64, device=dev, dtype=torch.float16))


@pytest.mark.parametrize("M,N,K", [(300, 320, 640), (1848, 768, 768), (8192, 640, 640), (32768, 320, 320)])
def test_bf16_linear(M, N, K):
    from textboost_amd import ops
    torch.manual_seed(0)
    A = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    ops.gemm(A, W, out, bias=bias, R=R)
    ref = A.float() @ W.float().T + bias + R.float()
    parity(f"bf16 linear {M}x{N}x{K}", out, ref, rel=4e-3, maxabs=1.6e-2)    # only the output rounding: 2^-9 relative per element


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 128, 16, 16), (8, 128, 320, 64, 64), (8, 640, 1280, 8, 8)])
def test_bf16_conv3x3_fwd_and_dgrad(B, Cin, Cout, H, W):
    from textboost_amd import ops
    from test_gpu_gemm import pack_conv_w, pack_conv_w_dgrad
    torch.manual_seed(1)
    x = torch.randn(B, Cin, H, W, device=dev).to(BF)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) / (3 * Cin ** 0.5)).to(BF)
    bias = torch.randn(Cout, device=dev)
    out = torch.empty(B * H * W, Cout, device=dev, dtype=BF)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cin, Hout=H, Wout=W, stride=1, sign=1, upsample=0, transposed=0)
    ops.gemm(nhwc(x).view(B * H * W, Cin), pack_conv_w(w), out, bias=bias, conv=geo)
    ref = nhwc(F.conv2d(x.float(), w.float(), bias, padding=1)).view(B * H * W, Cout)
    parity("bf16 conv3x3", out, ref, rel=4e-3, maxabs=1.6e-2)
    dy = torch.randn(B, Cout, H, W, device=dev).to(BF)
    dx = torch.empty(B * H * W, Cin, device=dev, dtype=BF)
    geo = dict(B=B, Hin=H, Win=W, Cin=Cout, Hout=H, Wout=W, stride=1, sign=-1, upsample=0, transposed=0)
    ops.gemm(nhwc(dy).view(B * H * W, Cout), pack_conv_w_dgrad(w), dx, conv=geo)
    refd = nhwc(F.conv_transpose2d(dy.float(), w.float(), padding=1)).view(B * H * W, Cin)
    parity("bf16 conv3x3 dgrad", dx, refd, rel=4e-3, maxabs=1.6e-2)


@pytest.mark.parametrize("B,H,Sq,Skv,hd", [(2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 77, 40), (3, 12, 77, 77, 64)])
def test_bf16_attention_fwd_bwd(B, H, Sq, Skv, hd):
    """the software-pipelined hd = 40 kernels (S = 4096), the LDS-DMA hd = 80 ones, the 77-key cross-attention and the encoder's short sequences"""
    from textboost_amd import ops
    torch.manual_seed(2)
    C = H * hd
    q = torch.randn(B * Sq, C, device=dev).to(BF)
    k = torch.randn(B * Skv, C, device=dev).to(BF)
    v = torch.randn(B * Skv, C, device=dev).to(BF)
    do = torch.randn(B * Sq, C, device=dev).to(BF)
    o = torch.empty_like(q)
    lse = torch.empty(B * H, Sq, device=dev)
    ops.attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd)
    qf, kf, vf = (t.float().view(B, -1, H, hd).transpose(1, 2).requires_grad_(True) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qf, kf, vf)
    ref.backward(do.float().view(B, Sq, H, hd).transpose(1, 2))
    parity("bf16 attention out", o.view(B, Sq, H, hd).transpose(1, 2), ref, rel=1.2e-2, maxabs=3e-2)   # P is rounded to bf16 before P.V
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B * H, Sq, device=dev)
    ws = torch.empty(16 * 2 * B * Skv, C, device=dev) if Skv == 77 and Sq > 77 else (torch.empty(2 * B * H, Sq, device=dev) if Sq % 128 == 0 else None)
    ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Sq, Skv, hd, ws=ws)
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        parity(f"bf16 attention {name}", got.view(B, -1, H, hd).transpose(1, 2), want, rel=2e-2, maxabs=5e-2)


def test_bf16_norms():
    from textboost_amd import ops
    torch.manual_seed(3)
    B, HW, C, G = 4, 1024, 640, 32
    x = torch.randn(B * HW, C, device=dev).to(BF)
    gam, bet = torch.randn(C, device=dev), torch.randn(C, device=dev)
    y = torch.empty_like(x)
    stats = torch.empty(B * G, 2, device=dev)
    ws = torch.empty(ops.groupnorm_ws(B, HW, C, G), device=dev)
    ops.groupnorm_fwd(x, y, gam, bet, stats, ws, B, HW, C, G, 1e-5, True)
    xr = x.float().view(B, HW, C).transpose(1, 2).requires_grad_(True)
    ref = F.silu(F.group_norm(xr, G, gam, bet, 1e-5))
    parity("bf16 groupnorm+silu", y.view(B, HW, C).transpose(1, 2), ref, rel=4e-3, maxabs=1.6e-2)
    dy = torch.randn(B * HW, C, device=dev).to(BF)
    ref.backward(dy.float().view(B, HW, C).transpose(1, 2))
    dx = torch.empty_like(x)
    ops.groupnorm_bwd(dy, x, gam, bet, stats, dx, ws, B, HW, C, G, True)
    parity("bf16 groupnorm backward", dx.view(B, HW, C).transpose(1, 2), xr.grad, rel=6e-3, maxabs=2e-2)
    M, D = 2048, 1280
    x = torch.randn(M, D, device=dev).to(BF)
    g2, b2 = torch.randn(D, device=dev), torch.randn(D, device=dev)
    y = torch.empty_like(x)
    st = torch.empty(M, 2, device=dev)
    ops.layernorm_fwd(x, y, g2, b2, st)
    parity("bf16 layernorm", y, F.layer_norm(x.float(), (D,), g2, b2), rel=4e-3, maxabs=1.6e-2)


def test_bf16_fused_feed_forward():
    from textboost_amd import ops
    from test_gpu_gemm import pack_geglu
    M, C, I = 384, 320, 1280
    torch.manual_seed(4)
    w1 = (torch.randn(2 * I, C, device=dev) / C ** 0.5).to(BF)
    b1 = torch.randn(2 * I, device=dev) * 0.3
    w2 = (torch.randn(C, I, device=dev) / I ** 0.5).to(BF)
    b2 = torch.randn(C, device=dev) * 0.3
    x = torch.randn(M, C, device=dev).to(BF)
    R = torch.randn(M, C, device=dev).to(BF)
    hg = torch.empty(M, 2 * I, device=dev, dtype=BF)
    y = torch.empty(M, C, device=dev, dtype=BF)
    ops.ff_fwd(x, pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous(), w2, b2, hg, y, R=R)
    proj = x.float() @ w1.float().T + b1
    h, g = proj[:, :I].to(BF).float(), proj[:, I:].to(BF).float()
    ref = (h * F.gelu(g)).to(BF).float() @ w2.float().T + b2 + R.float()
    parity("bf16 fused feed-forward", y, ref, rel=6e-3, maxabs=2e-2)
    dy = torch.randn(M, C, device=dev).to(BF)
    dx = torch.empty(M, C, device=dev, dtype=BF)
    ops.ff_bwd(dy, w2.t().contiguous(), pack_geglu(w1).t().contiguous(), hg, dx)
    blocks = hg.float().reshape(M, I // 32, 2, 32)
    hh = blocks[:, :, 0].reshape(M, I).requires_grad_(True)
    gg = blocks[:, :, 1].reshape(M, I).requires_grad_(True)
    (hh * F.gelu(gg)).backward(dy.float() @ w2.float())
    refd = hh.grad.to(BF).float() @ w1[:I].float() + gg.grad.to(BF).float() @ w1[I:].float()
    parity("bf16 fused feed-forward backward", dx, refd, rel=6e-3, maxabs=2e-2)


def test_bf16_row_chains_of_the_64x64_maps():
    """round 6: the chained launches (tb_chain320; tb_ff_fwd with pre_W / post_W) in the bfloat16 build -- the same sources, bf16 operands and bf16 LDS
    image -- against torch fp32 on the bf16-rounded intermediates (fp16 bounds x 8)."""
    from textboost_amd import ops
    from test_gpu_gemm import pack_geglu
    M, C, I = 384, 320, 1280
    torch.manual_seed(14)
    r = lambda *sh, s=1.0: torch.randn(*sh, device=dev) * s   # noqa: E731
    wa, ba, wb = (r(C, C) / C ** 0.5).to(BF), r(C, s=0.3), (r(960, C) / C ** 0.5).to(BF)
    gamma, beta = 1 + 0.4 * r(C), 0.2 * r(C)
    x, R = r(M, C).to(BF), (r(M, C) * 0.8 + r(M, 1) * 0.5).to(BF)
    t = torch.empty(M, C, device=dev, dtype=BF)
    y = torch.empty(M, 960, device=dev, dtype=BF)
    st = torch.zeros(M, 2, device=dev)
    ops.chain320(x, wa, ba, R, t, gamma, beta, st, wb, None, y)
    t_ref = x.float() @ wa.float().T + ba + R.float()
    th = t_ref.to(BF).float()
    l_ref = F.layer_norm(th, (C,), gamma, beta, 1e-5).to(BF).float()
    parity("bf16 chain320: t", t, t_ref, rel=4e-3, maxabs=1.6e-2)
    parity("bf16 chain320: y = LN(t) W2^T", y, l_ref @ wb.float().T, rel=1.2e-2, maxabs=4e-2)
    assert torch.allclose(st[:, 0], th.mean(1), rtol=1e-3, atol=2e-3)
    # the feed-forward with both neighbours
    w1, b1 = (r(2 * I, C) / C ** 0.5).to(BF), r(2 * I, s=0.3)
    w2, b2 = (r(C, I) / I ** 0.5).to(BF), r(C, s=0.3)
    wp, bp = (r(C, C) / C ** 0.5).to(BF), r(C, s=0.3)
    xin = r(M, C).to(BF)
    hg = torch.empty(M, 2 * I, device=dev, dtype=BF)
    t2 = torch.empty(M, C, device=dev, dtype=BF)
    out = torch.empty(M, C, device=dev, dtype=BF)
    ops.ff_fwd(x, pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous(), w2, b2, hg, None, R=t2,
               pre=(wa, ba, R, t2, gamma, beta, st, 1e-5), post=(wp, bp, xin, out))
    assert torch.equal(t2, t)                                   # the same stage, the same bits
    proj = l_ref @ w1.float().T + b1
    h, g = proj[:, :I].to(BF).float(), proj[:, I:].to(BF).float()
    t3 = ((h * F.gelu(g)).to(BF).float() @ w2.float().T + b2 + th).to(BF).float()
    parity("bf16 feed-forward chain: proj_out + block input", out, t3 @ wp.float().T + bp + xin.float(), rel=1.6e-2, maxabs=5e-2)


def test_bf16_unet_and_text_encoder_match_oracle():
    from test_gpu_model import make_unet, make_encoders, lora_grads_from_oracle
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    ref, hip, cfg = make_unet(B, hw, D)
    assert hip.dtype == BF
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, hw, hw, generator=g).to(BF).float()
    t = torch.tensor([17, 801])
    ehs = torch.randn(B, 77, D, generator=g).to(BF).float().requires_grad_(True)
    pred_ref = ref(x, t, ehs)
    dpred = torch.randn(B, 4, hw, hw, generator=g)
    pred_ref.backward(dpred)
    pred = hip.forward(x.to(BF).to(dev), t.to(dev), ehs.detach().to(BF).view(B * 77, D).to(dev).contiguous())
    # (the oracle's weights were rounded to fp16, the bf16 module rounds them once more: part of the mode's own error, as in the reference)
    parity("bf16 tiny UNet pred", pred, pred_ref, rel=2.4e-2, maxabs=3.2e-2, ch_dim=1, ch_rel=3.2e-2)
    d_ehs = hip.backward(dpred.to(dev))
    parity("bf16 tiny UNet d_ehs", d_ehs.view(B, 77, D), ehs.grad, rel=4e-2, maxabs=5e-2)
    student, teacher, enc, enc_teacher, added, null = make_encoders(3, D)
    ids = ts.synthetic_ids(3, added, g)
    out_ref = student(ids)
    Rr = torch.randn(3, 77, D, generator=g)
    (out_ref * Rr).sum().backward()
    enc.pack_lora()
    out = enc.forward(ids.to(dev), slot=0)
    parity("bf16 tiny encoder hidden states", out.view(3, 77, D), out_ref, rel=4e-3, maxabs=8e-3)
    enc.zero_grad()
    enc.backward(Rr.view(3 * 77, D).to(dev).contiguous(), slot=0)
    gA, gB = lora_grads_from_oracle(student)
    parity("bf16 tiny encoder grad lora_A", enc.grad_A, gA, rel=1.2e-2, maxabs=1.6e-2)
    parity("bf16 tiny encoder grad lora_B", enc.grad_B, gB, rel=1.2e-2, maxabs=1.6e-2)


def test_bf16_full_step_runs_without_a_grad_scaler_and_matches_oracle():
    from test_gpu_model import build_step
    from oracle import train_step as ts
    B, hw, D = 2, 16, 64
    st_ref, step, added = build_step(B, hw, D, use_scaler=False)
    g = torch.Generator().manual_seed(5)
    for it in range(2):
        ids, pids = ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True)
        x0, noise = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        out = st_ref.step(x0, noise, t, ids, pids)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t); step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        sc = step.scalars()
        assert sc["found_inf"] == 0.0 and sc["loss_scale"] == 1.0
        assert abs(sc["loss_mse"] - out["mse"]) < 5e-2 * abs(out["mse"]) + 1e-3, (sc, out["mse"])
        clip = min(1.0, 1.0 / (out["lora_grad_norm"] + 1e-6))
        gA = torch.stack([torch.cat(out["g_lora"][6 * l + 0: 6 * l + 6: 2]) for l in range(len(st_ref.te.layers))])
        parity(f"bf16 step {it} grad lora_A", step.te.grad_A * clip, gA, rel=3.2e-2, maxabs=5e-2)
        parity(f"bf16 step {it} grad added rows", step.te.grad_added, out["g_emb_added"], rel=3.2e-2, maxabs=5e-2)
    assert step.scalars()["opt_steps"] == 2.0
    # graph replay of the bf16 step == eager, bit for bit
    outs = []
    for mode in ("eager", "graph"):
        _, s2, added2 = build_step(B, hw, D, use_scaler=False)
        g2 = torch.Generator().manual_seed(6)
        s2.input_ids.copy_(ts.synthetic_ids(B, added2, g2)); s2.prior_ids.copy_(ts.synthetic_ids(B, added2, g2, prior=True))
        s2.x0.copy_(torch.randn(B, 4, hw, hw, generator=g2)); s2.noise.copy_(torch.randn(B, 4, hw, hw, generator=g2))
        s2.timesteps.copy_(torch.randint(0, 1000, (B,), generator=g2))
        if mode == "graph":
            s2.capture(warmup=2)
            s2.replay()
        else:
            for _ in range(3):
                s2.step_eager()
        torch.cuda.synchronize()
        outs.append((s2.te.lora_A.clone(), s2.te.token_table[49408:].clone()))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_bf16_unet_crossattn_kv_adapters_step_matches_oracle():
    """--unet_params_to_train crossattn_kv under --mixed_precision bf16 (train_textboost.py:712-721, :937: the reference trains the bf16-cast
    adapters; there is no GradScaler to refuse them -- round 6, VERDICT r5 "missing #3").  fp32 master adapters, bf16 operand copies in the hoisted
    K/V GEMM, gradients as exact-fp32 products on fp32 copies of the bf16 operands: two optimizer steps against the fp32 oracle at the bf16 bounds
    of the other step test (3.2e-2 / 5e-2); the adapters must move, and the step must differ from the one without adapters."""
    from oracle import train_step as ts
    from test_gpu_model import make_unet, make_encoders
    from textboost_amd.trainer import StepHyper, TextBoostStep
    B, hw, D, r = 2, 16, 64, 4
    ref_unet, hip_unet, _ = make_unet(B, hw, D, seed=3)
    adapters = ref_unet.add_crossattn_kv_adapters(r)
    hip_unet.enable_kv_lora(r, seed=0)
    with torch.no_grad():
        for ps in adapters.values():
            ps[1].normal_(std=0.05)
            ps[3].normal_(std=0.05)
    for l, (p, C) in enumerate(hip_unet.xattn):
        kA, kB, vA, vB = adapters[p]
        ko = hip_unet.kv_off[p]
        hip_unet.kv_lora_A[l, :r].copy_(kA.detach()); hip_unet.kv_lora_A[l, r:].copy_(vA.detach())
        hip_unet.kv_lora_B[ko:ko + C].copy_(kB.detach()); hip_unet.kv_lora_B[ko + C:ko + 2 * C].copy_(vB.detach())
    student, teacher, hip_te, hip_teacher, added, null = make_encoders(B, D, seed=4)
    unet_params = [q for ps in adapters.values() for q in ps]
    st_ref = ts.TrainState(student, teacher, ref_unet, added, ts.StepConfig(), unet_lora=unet_params)
    step = TextBoostStep(hip_unet, hip_te, hip_teacher, StepHyper(use_grad_scaler=False, init_scale=1.0), (B, 4, hw, hw), device=dev)
    assert step.n_unet == sum(q.numel() for q in unet_params) and hip_unet.kv_A16.dtype == BF
    step.external_noise = True
    g = torch.Generator().manual_seed(7)

    def flat(ps_list, which):
        A = torch.stack([torch.cat([ps_list[4 * l + 0], ps_list[4 * l + 2]]) for l in range(len(hip_unet.xattn))])
        Bm = torch.cat([torch.cat([ps_list[4 * l + 1], ps_list[4 * l + 3]]) for l in range(len(hip_unet.xattn))])
        return A if which == "A" else Bm

    A0 = hip_unet.kv_lora_A.clone()
    for it in range(2):
        ids, pids = ts.synthetic_ids(B, added, g), ts.synthetic_ids(B, added, g, prior=True)
        x0, noise = torch.randn(B, 4, hw, hw, generator=g), torch.randn(B, 4, hw, hw, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        out = st_ref.step(x0, noise, t, ids, pids)
        step.x0.copy_(x0); step.noise.copy_(noise); step.timesteps.copy_(t); step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        step.step_eager()
        sc = step.scalars()
        assert sc["found_inf"] == 0.0 and abs(sc["loss_mse"] - out["mse"]) < 5e-2 * abs(out["mse"]) + 1e-3
        parity(f"bf16 step {it} grad UNet lora_A", hip_unet.kv_grad_A, flat(out["g_unet"], "A"), rel=3.2e-2, maxabs=5e-2)
        parity(f"bf16 step {it} grad UNet lora_B", hip_unet.kv_grad_B, flat(out["g_unet"], "B"), rel=3.2e-2, maxabs=5e-2)
        parity(f"bf16 step {it} grad added rows (with the adapters' d_ehs term)", step.te.grad_added, out["g_emb_added"], rel=3.2e-2, maxabs=5e-2)
        cur = [q.detach() for q in unet_params]
        # Adam's first moves are ~lr * sign(g) (lr = 5e-5): an element whose gradient sits inside the bf16 noise may move the other way, 2 lr per step
        assert (hip_unet.kv_lora_A.cpu() - flat(cur, "A")).abs().max().item() < (it + 1) * 1.1e-4
        assert (hip_unet.kv_lora_B.cpu() - flat(cur, "B")).abs().max().item() < (it + 1) * 1.1e-4
    assert not torch.equal(hip_unet.kv_lora_A, A0)
    # the validation sampler's fold still works off the fp32 masters
    W = hip_unet.merged_kv_weight()
    assert W.dtype == torch.float32 and torch.isfinite(W).all()


def test_vae_encoder_stays_on_the_fp16_build_in_a_bf16_run():
    """train_textboost.py:938 keeps the VAE in fp32 in every --mixed_precision mode.  The device encoder multiplies 16-bit operands with fp32
    accumulation; in a bf16 run it must not drop to 8 significand bits: HipVAEEncoder packs and runs on the fp16 library whatever the process's
    active build is.  Latents against the fp32 oracle at the fp16 test's own bound (tests/test_gpu_vae.py: 5e-3 on this config), bit-equal to an
    fp16-process encoder, and measurably better than what bf16 operands give (the same encoder forced onto the bf16 build)."""
    from oracle.vae_encoder import VAEConfig, VAEEncoder
    from textboost_amd import _lib as L
    from textboost_amd import models
    from textboost_amd.vae import HipVAEEncoder, VAEGeometry, vae_encoder_shapes
    assert L.half_kind() == "bf16"
    geo = VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    B, H, W = 2, 64, 64
    sd = models.random_state_dict(vae_encoder_shapes(geo), 5, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    with torch.device("meta"):
        ref = VAEEncoder(VAEConfig.tiny())
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    eps = torch.randn(B, 4, H // 8, W // 8, generator=g)
    with torch.no_grad():
        z_ref = ref.encode_sample(x, noise=eps)
    dsd = {k: v.to(dev) for k, v in sd.items()}
    enc = HipVAEEncoder(geo, dsd, B, H, W, device=dev)
    assert enc.half == "fp16" and all(t.dtype != BF for t in enc.P.values())
    z = enc.encode(x.to(dev), noise=eps.to(dev)).clone()
    assert L.half_kind() == "bf16"                       # the switch is scoped to the encoder's own calls
    e16 = rel_err(z, z_ref)
    with L.use_half("fp16"):
        z16 = HipVAEEncoder(geo, dsd, B, H, W, device=dev).encode(x.to(dev), noise=eps.to(dev)).clone()
    assert torch.equal(z, z16)
    forced = HipVAEEncoder.__new__(HipVAEEncoder)        # the bf16-operand encoder this replaces (round 5's behaviour), for the comparison only
    forced.half = None
    nd = len(geo.block_out_channels) - 1
    forced.geo, forced.B, forced.H, forced.W, forced.dev, forced.dtype, forced._bufs, forced.generator = geo, B, H, W, dev, torch.float32, {}, None
    forced._pack(dsd)
    forced.gn_ws = torch.empty((2048 + 2 * B) * geo.norm_num_groups * 2, device=dev, dtype=torch.float32)
    ebf = rel_err(forced.encode(x.to(dev), noise=eps.to(dev)), z_ref)
    print(f"[parity] VAE latents in a bf16 run: fp16-build encoder {e16:.3e} vs the fp32 oracle (bf16 operands would give {ebf:.3e})")
    assert e16 < 5e-3 and ebf > 2 * e16


def test_bf16_cli_end_to_end(tmp_path):
    """`--mixed_precision bf16` through the CLI (synthetic latents, the full SD1.5 shapes at 16x16 latents): the reference output layout, no loss
    scaling, finite weights."""
    import sys
    from safetensors.torch import load_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    out = str(tmp_path / "run")
    T.main(T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", out, "--train_batch_size", "2",
                         "--resolution", "128", "--max_train_steps", "3", "--placeholder_token", "<dog>", "--lora_rank", "4",
                         "--mixed_precision", "bf16", "--learning_rate", "5e-5", "--emb_learning_rate", "1e-3", "--seed", "42"]))
    from textboost_amd import _lib as L
    assert L.half_kind() == "bf16"
    sd = load_file(os.path.join(out, "text_encoder", "adapter_model.safetensors"))
    assert len(sd) == 72 and any(v.abs().max() > 0 for k, v in sd.items() if "lora_B" in k)
    d = torch.load(os.path.join(out, "dog.bin"))
    assert d["<dog>"].shape == (768,) and torch.isfinite(d["<dog>"]).all()
    log = open(os.path.join(out, "training.log")).read()
    assert "bf16 mixed precision (no GradScaler)" in log
    # ... and with the UNet's cross-attention K/V adapters (round 6): runs, writes <out>/unet/ in peft's adapter layout
    out2 = str(tmp_path / "run_kv")
    T.main(T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--output_dir", out2, "--train_batch_size", "2",
                         "--resolution", "128", "--max_train_steps", "2", "--placeholder_token", "<dog>", "--lora_rank", "4",
                         "--mixed_precision", "bf16", "--unet_params_to_train", "crossattn_kv", "--seed", "42"]))
    sdu = load_file(os.path.join(out2, "unet", "adapter_model.safetensors"))
    assert len(sdu) == 64 and any(v.abs().max() > 0 for k, v in sdu.items() if "lora_B" in k)
