"""A/B + correctness of the software-pipelined attention kernels (attention_il.hip) against the LDS-DMA kernels and fp32 torch SDPA."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def ref_attn(q, k, v, B, H, S, hd):
    qf, kf, vf = (t.float().view(B, S, H, hd).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * hd ** -0.5
    lse = torch.logsumexp(s, -1)
    o = torch.softmax(s, -1) @ vf
    return o.transpose(1, 2).reshape(B * S, H * hd), lse
def check(B, H, S, hd, spike=False, seed=0):
    torch.manual_seed(seed)
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device=dev).half()
    if spike:  # force the re-base branch: a few keys late in the sequence score far above everything before them
        qkv = qkv.clone()
        for b in range(B):
            for h in range(H):
                kk = S // 2 + 7 + 8 * h
                qrow = qkv[b * S + 5, h * hd:(h + 1) * hd].float()
                qkv[b * S + kk, C + h * hd:C + (h + 1) * hd] = (qrow * 6).half()
                qkv[b * S + kk + 70, C + h * hd:C + (h + 1) * hd] = (qkv[b * S + 300, h * hd:(h + 1) * hd].float() * 12).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
    oref, lref = ref_attn(q, k, v, B, H, S, hd)
    res = {}
    for name, var in (("il", 1), ("dma", 1 | 1024)):
        L.lib().tb_attention_set_variant(var)
        o.zero_(); lse.zero_()
        ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
        torch.cuda.synchronize()
        e = ((o.float() - oref).norm() / oref.norm()).item()
        em = ((o.float() - oref).abs().max() / oref.abs().max()).item()
        el = (lse - lref).abs().max().item()
        res[name] = (e, em, el)
    L.lib().tb_attention_set_variant(1)
    print(f"B={B} H={H} S={S} hd={hd} spike={spike}: " + "  ".join(f"{n}: relL2 {a:.2e} maxabs {b_:.2e} lse {c:.2e}" for n, (a, b_, c) in res.items()))
    tol_lse = 3e-2 if spike else 2e-3  # spiked scores ~40-80 nats: the fp16 rounding of the scaled Q operand alone is ~2e-2 there
    assert res["il"][0] < 2e-3 and res["il"][1] < 4e-3 and res["il"][2] < tol_lse, res
if __name__ == "__main__":
    check(1, 8, 512, 40)
    check(2, 8, 1024, 40, spike=True)
    check(8, 8, 4096, 40)
    check(1, 8, 4096, 40, spike=True, seed=3)
    B, H, S, hd = 8, 8, 4096, 40; C = H * hd
    qkv = torch.randn(B * S, 3 * C, device=dev).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
    fl = 4 * B * H * S * S * hd
    for rnd in range(3):
        for name, var in (("il", 1), ("dma", 1 | 1024)):
            L.lib().tb_attention_set_variant(var)
            t = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
            print(f"round {rnd} {name:4s}: fwd {t:7.1f} us  {fl/t/1e6:6.1f} TF/s")
    L.lib().tb_attention_set_variant(1)

    # ---- backward: software-pipelined dK/dV (and dQ) kernels vs the LDS-DMA ones and fp32 autograd
    def bwd_check(B, H, S, hd=40, seed=1):
        torch.manual_seed(seed)
        C = H * hd
        qkv = torch.randn(B * S, 3 * C, device=dev).half()
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
        L.lib().tb_attention_set_variant(1)
        ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
        do = torch.randn(B * S, C, device=dev).half()
        qr, kr, vr = [t.float().view(B, S, H, hd).transpose(1, 2).requires_grad_(True) for t in (q, k, v)]
        sc = qr @ kr.transpose(-1, -2) * hd ** -0.5
        (torch.softmax(sc, -1) @ vr).backward(do.float().view(B, S, H, hd).transpose(1, 2))
        refs = [t.grad.transpose(1, 2).reshape(B * S, C) for t in (qr, kr, vr)]
        for name, var in (("il", 1), ("dma", 1 | 2048 | 4096)):
            L.lib().tb_attention_set_variant(var)
            delta = torch.empty(B, H, S, device=dev)
            dqkv = torch.zeros(B * S, 3 * C, device=dev, dtype=torch.float16)
            ws = torch.empty(2 * B * H * S, device=dev)
            ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
            torch.cuda.synchronize()
            errs = [((dqkv[:, i * C:(i + 1) * C].float() - refs[i]).norm() / refs[i].norm()).item() for i in range(3)]
            print(f"bwd B={B} H={H} S={S} {name}: dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}")
            assert max(errs) < 4e-3, errs
        L.lib().tb_attention_set_variant(1)
    bwd_check(8, 8, 512)
    bwd_check(1, 8, 4096)
    B, H, S, hd = 8, 8, 4096, 40; C = H * hd
    do = torch.randn(B * S, C, device=dev).half(); delta = torch.empty(B, H, S, device=dev)
    dqkv = torch.zeros(B * S, 3 * C, device=dev, dtype=torch.float16); ws = torch.empty(2 * B * H * S, device=dev)
    for rnd in range(3):
        for name, var in (("il", 1), ("dma", 1 | 2048 | 4096)):
            L.lib().tb_attention_set_variant(var)
            t = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws))
            print(f"round {rnd} {name:4s}: bwd {t:7.1f} us  {2.5*fl/t/1e6:6.1f} TF/s")
    L.lib().tb_attention_set_variant(1)
