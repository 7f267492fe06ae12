"""text-encoder Linear shapes (M = 1848 rows) on the 4-wave kernel vs the ragged 128x128 8-wave tile (tb_gemm8_set bit 131072)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib(); dev = "cuda"
def graph_time(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3
torch.manual_seed(0)
base = lib.tb_gemm8_set(39); lib.tb_gemm8_set(base)
for M in (1848, 1232):
    for N, K, act, c2 in [(3072, 768, L.ACT_QUICK_GELU, True), (3072, 768, L.ACT_QUICK_GELU_GRAD, True), (2304, 768, L.ACT_NONE, False), (3072, 768, L.ACT_NONE, False)]:
        A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half(); bias = torch.randn(N, device=dev)
        pre = torch.randn(M, N, device=dev).half() if c2 else None
        outs, ts = [], []
        for bits in (base, base | 131072):
            lib.tb_gemm8_set(bits)
            out = torch.zeros(M, N, device=dev, dtype=torch.half)
            f = lambda: ops.gemm(A, W, out, bias=None if act == L.ACT_QUICK_GELU_GRAD else bias, act=act, C2=pre)
            ts.append(graph_time(f)); outs.append(out.clone())
        lib.tb_gemm8_set(base)
        ref = A.float() @ W.float().t()
        err = (outs[0].float() - outs[1].float()).abs().max().item()
        print(f"M={M} N={N} K={K} act={act}: 4-wave {ts[0]:6.1f} us  8-wave 128x128 {ts[1]:6.1f} us   max|diff| {err:.3e}  ({2*M*N*K/ts[1]/1e6:.0f} TFLOP/s)")
