import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; S, hd, B, H = 4096, 40, 8, 8; C = H * hd
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
for fill in ("randn", "zeros", "randn*0.1", "const0.5"):
    if fill == "randn": qkv = torch.randn(B * S, 3 * C, device=dev).half()
    elif fill == "zeros": qkv = torch.zeros(B * S, 3 * C, device=dev).half()
    elif fill == "randn*0.1": qkv = (torch.randn(B * S, 3 * C, device=dev) * 0.1).half()
    else: qkv = torch.full((B * S, 3 * C), 0.5, device=dev).half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    for name, var in (("il", 1), ("dma", 1 | 1024)):
        L.lib().tb_attention_set_variant(var)
        t = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
        print(f"{fill:10s} {name:4s}: fwd {t:7.1f} us")
