"""Deep LDS ring (1 block/CU, 4-6 stages) vs the default 2-stage / 2-3 blocks per CU, per shape (graph-captured, rotating buffers)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(M, N, K, nbuf=6, reps=5):
    As = [torch.randn(M, K, device=dev).half() for _ in range(nbuf)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nbuf)]
    Rs = [torch.randn(M, N, device=dev).half() for _ in range(nbuf)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    bias = torch.randn(N, device=dev)
    def run():
        for i in range(nbuf): ops.gemm(As[i], Ws[i], Os[i], bias=bias, R=Rs[i])
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * nbuf) * 1e3
lib = L.lib(); lib.tb_gemm_set_variant(1000)
for M, N, K in [(8192, 640, 640), (32768, 320, 320), (8192, 640, 2560), (32768, 320, 2560), (8192, 640, 5120), (32768, 1280, 320), (8192, 2560, 640), (2048, 5120, 1280), (2048, 3840, 1280), (32768, 320, 1280)]:
    out = []
    for ft in (2, 3):
        for v in (0, 4, 5, 6):
            lib.tb_gemm_set_variant(8000 + ft); lib.tb_gemm_set_variant(v); out.append(bench(M, N, K))
    lib.tb_gemm_set_variant(8000); lib.tb_gemm_set_variant(0)
    print(f"M={M:6d} N={N:5d} K={K:5d}: 128x64 x2 {out[0]:6.1f} x3 {out[1]:6.1f} x4 {out[2]:6.1f} x6 {out[3]:6.1f} | 128x128 x2 {out[4]:6.1f} x3 {out[5]:6.1f} x4 {out[6]:6.1f}", flush=True)
