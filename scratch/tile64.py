"""k-tile depth of the 64x64 tile on the shapes that use it (graph-captured, rotating buffers, residual epilogue)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(M, N, K, resid=True, nbuf=6, reps=5):
    As = [torch.randn(M, K, device=dev).half() for _ in range(nbuf)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nbuf)]
    Rs = [torch.randn(M, N, device=dev).half() for _ in range(nbuf)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    bias = torch.randn(N, device=dev)
    def run():
        for i in range(nbuf):
            ops.gemm(As[i], Ws[i], Os[i], bias=bias, R=Rs[i] if resid else None)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * nbuf) * 1e3
lib = L.lib()
for M, N, K in [(2048, 1280, 1280), (512, 1280, 1280), (2048, 1280, 3840), (2048, 1280, 5120), (1232, 768, 768), (1232, 3072, 768), (1232, 768, 3072), (1232, 2304, 832), (616, 768, 768), (616, 3072, 768), (616, 768, 3072), (616, 2304, 768), (2048, 640, 1280), (512, 1280, 2560), (2048, 1280, 2560)]:
    r = []
    for v in (10, 14, 15):
        lib.tb_gemm_set_variant(v); r.append(bench(M, N, K))
    lib.tb_gemm_set_variant(1000)
    ns = []
    for v in (10, 14, 15):
        lib.tb_gemm_set_variant(v); ns.append(bench(M, N, K))
    lib.tb_gemm_set_variant(1384); lib.tb_gemm_set_variant(15)
    print(f"M={M:5d} N={N:5d} K={K:5d}: auto(split) x2 {r[0]:6.1f} x3 {r[1]:6.1f} x4 {r[2]:6.1f} | nosplit x2 {ns[0]:6.1f} x3 {ns[1]:6.1f} x4 {ns[2]:6.1f}", flush=True)
