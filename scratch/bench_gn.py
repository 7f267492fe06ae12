import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for HW, C in [(4096, 320), (4096, 640), (4096, 960), (1024, 640), (1024, 1280), (1024, 1920), (256, 1280), (256, 2560), (64, 1280), (64, 2560)]:
    B = 8; M = B * HW
    x = torch.randn(M, C, device=dev).half(); y = torch.empty_like(x); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    st = torch.empty(B, 32, 2, device=dev); ws = torch.empty(ops.groupnorm_ws(B, HW, C), device=dev); dy = torch.randn_like(x); dx = torch.empty_like(x)
    t = timeit(lambda: ops.groupnorm_fwd(x, y, g, b, st, ws, B, HW, C, silu=True))
    tb = timeit(lambda: ops.groupnorm_bwd(dy, x, g, b, st, dx, ws, B, HW, C, silu=True))
    print(f"  HW={HW:5d} C={C:5d}: fwd {t*1e6:7.1f} us ({3*M*C*2/t/1e9:6.0f} GB/s alg)  bwd {tb*1e6:7.1f} us ({5*M*C*2/tb/1e9:6.0f} GB/s alg)")
