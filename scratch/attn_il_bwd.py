import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def bwd_check(B, H, S, hd=40, seed=1):
    torch.manual_seed(seed)
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device=dev).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
    L.lib().tb_attention_set_variant(1)
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    do = torch.randn(B * S, C, device=dev).half()
    res = {}
    for name, var in (("il", 1 | 4096), ("dma", 1 | 2048), ("reg", 1 | 128 | 256)):
        L.lib().tb_attention_set_variant(var)
        delta = torch.empty(B, H, S, device=dev)
        dqkv = torch.zeros(B * S, 3 * C, device=dev, dtype=torch.float16)
        ws = torch.empty(2 * B * H * S, device=dev)
        ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
        torch.cuda.synchronize()
        res[name] = dqkv.float()
    L.lib().tb_attention_set_variant(1)
    ref = res["reg"]
    for name in ("il", "dma"):
        errs = [((res[name][:, i * C:(i + 1) * C] - ref[:, i * C:(i + 1) * C]).norm() / ref[:, i * C:(i + 1) * C].norm()).item() for i in range(3)]
        print(f"bwd B={B} H={H} S={S} {name} vs register-staged kernels: dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}  bit-equal to dma: {torch.equal(res[name], res['dma'])}")
        assert max(errs) < 2e-3 and torch.isfinite(res[name]).all()
bwd_check(2, 8, 4096)
bwd_check(8, 8, 1024)
bwd_check(8, 8, 256 * 3)
B, H, S, hd = 8, 8, 4096, 40; C = H * hd
for fill in ("randn", "zeros"):
    qkv = (torch.randn(B * S, 3 * C, device=dev) if fill == "randn" else torch.zeros(B * S, 3 * C, device=dev)).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    do = (torch.randn(B * S, C, device=dev) if fill == "randn" else torch.zeros(B * S, C, device=dev)).half(); delta = torch.empty(B, H, S, device=dev)
    dqkv = torch.zeros(B * S, 3 * C, device=dev, dtype=torch.float16); ws = torch.empty(2 * B * H * S, device=dev)
    for rnd in range(2):
        for name, var in (("il", 1 | 4096), ("dma", 1 | 2048)):
            L.lib().tb_attention_set_variant(var)
            t = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws))
            print(f"{fill} round {rnd} {name:4s}: bwd {t:7.1f} us")
L.lib().tb_attention_set_variant(1)
