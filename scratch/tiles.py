"""Per-shape tile choice for linear GEMMs: 64x64 vs 128x64 vs 128x128 (graph-captured, rotating buffers, residual epilogue)."""
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
shapes = [(8192, 640, 640), (32768, 320, 320), (2048, 1280, 1280), (512, 1280, 1280), (8192, 640, 2560), (32768, 320, 1280), (32768, 320, 2560), (8192, 640, 5120),
          (32768, 960, 320), (8192, 1920, 640), (2048, 3840, 1280), (32768, 1280, 320), (8192, 2560, 640), (2048, 5120, 1280), (2048, 1280, 5120), (2048, 1280, 10240),
          (8192, 640, 1920), (32768, 320, 960), (2048, 1280, 3840), (1232, 768, 768), (1232, 3072, 768), (1232, 768, 3072), (1232, 2304, 832), (616, 768, 768), (616, 3072, 768), (616, 768, 3072)]
lib = L.lib()
lib.tb_gemm_set_variant(1000)   # no split-K: tile shape / k-tile depth only
for M, N, K in shapes:
    res = {}
    for ft, tn in ((1, "64x64"), (2, "128x64"), (3, "128x128")):
        for v, vn in ((0, "k64x2"), (1, "k32x3"), (2, "k32x2"), (3, "k32x4")):
            lib.tb_gemm_set_variant(8000 + ft); lib.tb_gemm_set_variant(v)
            res[(tn, vn)] = bench(M, N, K)
    lib.tb_gemm_set_variant(8000); lib.tb_gemm_set_variant(0)
    auto = bench(M, N, K)
    best = min(res, key=res.get)
    print(f"M={M:6d} N={N:5d} K={K:5d}: auto {auto:6.1f} | " + " | ".join(f"{tn}: " + " ".join(f"{res[(tn, vn)]:6.1f}" for vn in ("k64x2", "k32x3", "k32x2", "k32x4")) for tn in ("64x64", "128x64", "128x128")) + f" | best {best[0]} {best[1]} {res[best]/auto:.2f}", flush=True)
