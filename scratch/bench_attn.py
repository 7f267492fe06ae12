import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for S, Skv, hd, H, causal in [(4096, 4096, 40, 8, False), (1024, 1024, 80, 8, False), (256, 256, 160, 8, False), (4096, 77, 40, 8, False), (1024, 77, 80, 8, False), (77, 77, 64, 12, True)]:
    B = 8 if not causal else 16; C = H * hd
    q = torch.randn(B * S, C, device=dev).half(); k = torch.randn(B * Skv, C, device=dev).half(); v = torch.randn(B * Skv, C, device=dev).half()
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev); do = torch.randn_like(q); delta = torch.empty_like(lse)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    t = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, Skv, hd, causal=causal))
    fl = 4 * B * H * S * Skv * hd
    tb = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, S, Skv, hd, causal=causal))
    print(f"  S={S:5d} Skv={Skv:5d} hd={hd:3d}: fwd {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s | bwd {tb*1e6:8.1f} us {2.5*fl/tb/1e12:6.1f} TF/s(alg 2.5x)")
