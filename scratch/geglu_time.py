"""GEGLU projection / GEGLU-backward launches at the 32x32 and 16x16 maps and the 128x320 Linear tile, library by TB_LIB_SUFFIX (round 5 DMAC2 A/B)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
NR = 6
def timeit(fn, reps=12):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NR)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
out = []
for M, C in ((8192, 640), (2048, 1280)):
    A = [torch.randn(M, C, device="cuda").half() for _ in range(NR)]
    W = [(torch.randn(8 * C, C, device="cuda") / C ** 0.5).half() for _ in range(NR)]
    b = torch.randn(8 * C, device="cuda")
    gated = [torch.empty(M, 4 * C, device="cuda", dtype=torch.float16) for _ in range(NR)]
    raw = [torch.empty(M, 8 * C, device="cuda", dtype=torch.float16) for _ in range(NR)]
    out.append(f"GEGLU {M}x{8*C}x{C}: {timeit(lambda i: ops.gemm(A[i], W[i], gated[i], bias=b, act=L.ACT_GEGLU, C2=raw[i])):.1f}")
    W2 = [(torch.randn(4 * C, C, device="cuda") / C ** 0.5).half() for _ in range(NR)]
    dproj = [torch.empty(M, 8 * C, device="cuda", dtype=torch.float16) for _ in range(NR)]
    out.append(f"GEGLU_GRAD {M}x{4*C}x{C}: {timeit(lambda i: ops.gemm(A[i], W2[i], dproj[i], act=L.ACT_GEGLU_GRAD, C2=raw[i])):.1f}")
for M, N, K in ((32768, 320, 1280), (32768, 960, 320), (32768, 320, 960)):
    A = [torch.randn(M, K, device="cuda").half() for _ in range(NR)]
    W = [(torch.randn(N, K, device="cuda") / K ** 0.5).half() for _ in range(NR)]
    o = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(NR)]
    out.append(f"{M}x{N}x{K}: {timeit(lambda i: ops.gemm(A[i], W[i], o[i])):.1f}")
print(os.environ.get("TB_LIB_SUFFIX", "base"), " | ".join(out))
