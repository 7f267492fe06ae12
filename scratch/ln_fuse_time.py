"""fused LayerNorm epilogues vs the two launches they replace, isolated (graph of 20 repeats, rotating buffers so nothing stays in L2 by accident)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
M, N = 32768, 320
NB = 6
def bufs(*shape, dtype=torch.float16):
    return [torch.randn(*shape, device="cuda").to(dtype) for _ in range(NB)]
def timeit(name, fn, reps=24):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NB)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / (2 * reps) * 1e3
    print(f"{name:60s} {us:8.2f} us", flush=True)
    return us
gamma, beta = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
bias = torch.randn(N, device="cuda")
for K in (320, 960, 2560):
    A, W = bufs(M, K), (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    R, T, Y, X = bufs(M, N), bufs(M, N), bufs(M, N), bufs(M, N)
    ST = bufs(M, 2, dtype=torch.float32)
    if K == 320:
        a = timeit(f"K={K} fwd: gemm(+bias+R)", lambda i: ops.gemm(A[i], W, T[i], bias=bias, R=R[i]))
        b = timeit(f"K={K} fwd: layernorm_fwd", lambda i: ops.layernorm_fwd(T[i], Y[i], gamma, beta, ST[i]))
        c = timeit(f"K={K} fwd: fused", lambda i: ops.gemm(A[i], W, T[i], bias=bias, R=R[i], ln_fwd=(gamma, beta, ST[i], Y[i], 1e-5)))
        print(f"   -> {a + b:.1f} us vs fused {c:.1f} us")
    a = timeit(f"K={K} bwd: gemm", lambda i: ops.gemm(A[i], W, T[i]))
    b = timeit(f"K={K} bwd: layernorm_bwd(+add)", lambda i: ops.layernorm_bwd(T[i], X[i], gamma, ST[i], Y[i], add=R[i]))
    c = timeit(f"K={K} bwd: fused", lambda i: ops.gemm(A[i], W, Y[i], R=R[i], ln_bwd=(gamma, ST[i], X[i])))
    print(f"   -> {a + b:.1f} us vs fused {c:.1f} us")
