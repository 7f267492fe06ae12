"""Does a cold (HBM) weight operand slow the GEMM down vs an L2/MALL-resident one?  A and C always rotate through > 256 MB."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def bench(M, N, K, nA, nW, reps=3):
    As = [torch.randn(M, K, device=dev).half() for _ in range(nA)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nW)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nA)]
    n = max(nA, nW)
    def run():
        for i in range(n): ops.gemm(As[i % nA], Ws[i % nW], Os[i % nA])
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * n) * 1e3
for M, N, K in [(8192, 640, 2560), (8192, 640, 640), (2048, 1280, 5120), (32768, 320, 1280)]:
    wb = N * K * 2 / 1e6; ab = (M * K + M * N) * 2 / 1e6
    nA = max(2, int(600 / ab) + 1)            # A + C cycle through ~600 MB: always cold-ish
    hot = bench(M, N, K, nA, 1)
    cold = bench(M, N, K, nA, max(2, int(600 / wb) + 1) if wb * 400 < 20000 else 64)
    a_hot = bench(M, N, K, 1, 1)
    print(f"M={M} N={N} K={K} (W {wb:.1f} MB, A+C {ab:.1f} MB): all hot {a_hot:6.1f} us | A cold, W hot {hot:6.1f} us | A cold, W cold {cold:6.1f} us", flush=True)
