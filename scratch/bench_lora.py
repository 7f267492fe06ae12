import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n // 10): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n // 10 * 10) * 1e3
for M, D, r in [(1232, 768, 4), (1232, 1024, 8)]:
    P = 3
    x = torch.randn(M, D, device=dev).half(); A = torch.randn(P * r, D, device=dev) / r; Bc = torch.randn(P * D, r, device=dev) * 0.1
    t = torch.zeros(M, 64, device=dev, dtype=torch.float16); ops.lora_down(x, A, t)
    dY = torch.randn(M, P * D, device=dev).half(); dt = torch.zeros(M, 64, device=dev, dtype=torch.float16)
    dA = torch.zeros_like(A); dB = torch.zeros_like(Bc)
    print(f"lora_bwd M={M} D={D} r={r}: {timeit(lambda: ops.lora_bwd(dY, x, t, Bc, dt, dA, dB, D, D, r, P)):.1f} us (fused + reduce)")
