"""Short-K linear GEMMs (attention projections): where do the ~38 us go?  Graph-captured, rotating buffers, ablations."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(M, N, K, resid, nbuf=8, reps=5):
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
shapes = [(8192, 640, 640), (32768, 320, 320), (2048, 1280, 1280), (8192, 640, 2560), (32768, 320, 1280)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in sys.argv[1].split(","))]
for M, N, K in shapes:
    r = []
    for abl in (0, 1, 2, 3, 7):
        L.lib().tb_gemm_set_variant(2000 + abl)
        r.append(bench(M, N, K, True))
    L.lib().tb_gemm_set_variant(2000)
    nr = bench(M, N, K, False)
    print(f"M={M} N={N} K={K}: full {r[0]:6.1f} us ({2*M*N*K/r[0]/1e6:6.1f} TF/s) | no-loads {r[1]:6.1f} | no-mfma {r[2]:6.1f} | neither {r[3]:6.1f} | +no-epi {r[4]:6.1f} | full w/o residual {nr:6.1f}", flush=True)
