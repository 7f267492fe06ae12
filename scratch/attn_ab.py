import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for S, hd in [(4096, 40)]:
    B, H = 8, 8; C = H * hd
    qkv = torch.randn(B * S, 3 * C, device=dev).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
    fl = 4 * B * H * S * S * hd
    do = torch.randn(B * S, C, device=dev).half(); delta = torch.empty(B, H, S, device=dev)
    dq = torch.empty(B * S, C, device=dev, dtype=torch.float16); dk = torch.empty_like(dq); dv = torch.empty_like(dq)
    for rnd in range(2):
        for var in (1, 33, 0):
            L.lib().tb_attention_set_variant(var)
            t = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
            tb = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, S, S, hd))
            print(f"S={S} hd={hd} variant={var}: fwd {t:7.1f} us  {fl/t/1e6:6.1f} TF/s | bwd {tb:7.1f} us {2.5*fl/tb/1e6:6.1f} TF/s")
