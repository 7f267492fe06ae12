"""Text-encoder GEMM shapes (cold weights, graph of rotating launches): split-K on / off, tile overrides.  usage: te_gemm_ab.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib()
dev = "cuda"
def bench(M, N, K, codes, reset, reps=5, res=False):
    wb = N * K * 2 / 1e6
    nW = min(200, max(4, int(700 / wb) + 1))
    As = [torch.randn(M, K, device=dev).half() for _ in range(3)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nW)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(3)]
    R = torch.randn(M, N, device=dev) if res else None
    for c in codes: lib.tb_gemm_set_variant(c)
    def run():
        for i in range(nW): ops.gemm(As[i % 3], Ws[i], Os[i % 3], R=R)
    run(); torch.cuda.synchronize()
    cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    for c in reset: lib.tb_gemm_set_variant(c)
    return s.elapsed_time(e) / (reps * nW) * 1e3, f"{cfg[0]}x{cfg[1]} st{cfg[3] % 10} S{cfg[4]}"
variants = [("default", [], []), ("no split", [1000], [1384]), ("no split 64x64x4st", [1000, 8001, 15], [1384, 8000, 9]),
            ("no split 128x64", [1000, 8002], [1384, 8000]), ("no split 64x64x2st", [1000, 8001, 10], [1384, 8000, 9]),
            ("split target 256", [1256], [1384]), ("split min 4 tiles", [4004], [4008])]
shapes = [(1848, 768, 3072), (1232, 768, 3072), (1232, 3072, 768), (1232, 768, 2368), (1848, 768, 768), (1232, 2304, 768), (2048, 1280, 5120), (2048, 1280, 1280), (512, 1280, 5120), (512, 1280, 1280)]
for M, N, K in shapes:
    print(f"{M}x{N}x{K}", flush=True)
    for name, codes, reset in variants:
        t, cfg = bench(M, N, K, codes, reset, res=True)
        print(f"    {name:22s} {t:7.1f} us   {cfg}", flush=True)
