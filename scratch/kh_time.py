"""round 5: the 128 x 80 one-per-CU Linear tile as 4 x 1 x 2 k-halves against the 8 x 1 waves (tb_gemm8_set bit 16384), cold weights (rotating operands)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
NR = int(os.environ.get('NR', '12'))
def timeit(fn, reps=24):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NR)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
SH = [tuple(int(v) for v in a.split("x")) for a in os.environ.get("SHAPES", "2048x1280x1280,2048x1280x640,2048x1280x2560,2048x1280x1920").split(",")]
for M, N, K in SH:
    A = [torch.randn(M, K, device="cuda").half() for _ in range(NR)]
    W = [(torch.randn(N, K, device="cuda") / K ** 0.5).half() for _ in range(NR)]
    R = [torch.randn(M, N, device="cuda").half() for _ in range(NR)]
    out = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(NR)]
    res = []
    for bits in (39, 39 | 16384, 39, 39 | 16384):
        L.lib().tb_gemm8_set(bits)
        res.append(timeit(lambda i: ops.gemm(A[i], W[i], out[i], R=R[i])))
    L.lib().tb_gemm8_set(39)
    print(f"M={M} N={N} K={K}: k-halves {res[0]:.1f} / {res[2]:.1f} us, 8x1 waves {res[1]:.1f} / {res[3]:.1f} us", flush=True)
