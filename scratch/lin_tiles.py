"""128x320 (NS=2) vs 64x320 (NS=3) Linear tiles per K at the 64x64-map shapes (M = 32768, N = 320) and 32x32 (M = 8192, N = 640)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
def timeit(fn, reps=20):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % 4)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
for M, N in ((32768, 320),):
    for K in (320, 640, 960, 1280, 2560):
        A = [torch.randn(M, K, device="cuda").half() for _ in range(4)]
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        R = [torch.randn(M, N, device="cuda").half() for _ in range(4)]
        out = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(4)]
        res = []
        for bits in (39, 39 | 8):
            L.lib().tb_gemm8_set(bits)
            res.append(timeit(lambda i: ops.gemm(A[i], W, out[i], R=R[i])))
        L.lib().tb_gemm8_set(39)
        print(f"M={M} N={N} K={K}: 128x320 {res[0]:.1f} us, 64x320 {res[1]:.1f} us", flush=True)
