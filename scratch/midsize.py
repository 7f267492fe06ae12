import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(M, N, K, nbuf=24, res=True):
    As = [torch.randn(M, K, device=dev).half() for _ in range(nbuf)]
    Ws = [torch.randn(N, K, device=dev).half() for _ in range(nbuf)]
    Rs = [torch.randn(M, N, device=dev).half() for _ in range(nbuf)]
    outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    bias = torch.randn(N, device=dev)
    def run():
        for i in range(nbuf): ops.gemm(As[i], Ws[i], outs[i], bias=bias, R=Rs[i] if res else None)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 / nbuf * 1e3
for shape in [(8192, 640, 640), (32768, 320, 320), (2048, 1280, 1280), (1232, 3072, 768), (1232, 768, 3072)]:
    row = []
    for v, abl in [(0, 0), (1, 0), (3, 0), (4, 0), (0, 1), (0, 2), (0, 3), (0, 7)]:
        L.lib().tb_gemm_set_variant(v); L.lib().tb_gemm_set_variant(2000 + abl)
        row.append(bench(*shape))
    L.lib().tb_gemm_set_variant(0); L.lib().tb_gemm_set_variant(2000)
    print(shape, "us: v0 %.1f | v1(BK32x3) %.1f | v3(BK32x4) %.1f | v4(BK64x3) %.1f || v0 no-loads %.1f | no-mfma %.1f | neither %.1f | +no-epi %.1f" % tuple(row))
