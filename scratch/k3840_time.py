import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
NR = 8
def timeit(fn, reps=24):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NR)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
for M, N, K in [(2048, 1280, 3840), (8192, 640, 5120), (8192, 640, 3840)]:
    A = [torch.randn(M, K, device="cuda").half() for _ in range(NR)]
    W = [(torch.randn(N, K, device="cuda") / K ** 0.5).half() for _ in range(NR)]
    out = [torch.empty(M, N, device="cuda", dtype=torch.float16) for _ in range(NR)]
    res = []
    for bits in (39 | 262144, 39, 39 | 262144, 39):
        L.lib().tb_gemm8_set(bits)
        res.append(timeit(lambda i: ops.gemm(A[i], W[i], out[i])))
    L.lib().tb_gemm8_set(39)
    print(f"M={M} N={N} K={K}: old route {res[0]:.1f} / {res[2]:.1f} us, one-per-CU 8-wave {res[1]:.1f} / {res[3]:.1f} us  max diff {(out[0]-out[1]).abs().max().item()}", flush=True)
