import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(M, N, K, nbuf=4, reps=5):
    As = [torch.randn(M, K, device=dev).half() for _ in range(nbuf)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nbuf)]
    Os = [torch.empty(M, N // 2, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    C2 = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    bias = torch.randn(N, device=dev)
    def run():
        for i in range(nbuf): ops.gemm(As[i], Ws[i], Os[i], bias=bias, act=L.ACT_GEGLU, C2=C2[i])
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * nbuf) * 1e3
for M, N, K in [(32768, 2560, 320), (8192, 5120, 640), (2048, 10240, 1280)]:
    L.lib().tb_gemm_set_variant(8004); a = bench(M, N, K)
    L.lib().tb_gemm_set_variant(8000); b = bench(M, N, K)
    print(f"GEGLU M={M} N={N} K={K}: 128x128 {a:6.1f} us, rule(128x64) {b:6.1f} us")
