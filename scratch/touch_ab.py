"""Touch-prefetch A/B on cold weights: a graph of launches that rotate through > 600 MB of distinct weight matrices (HBM-cold, as in the step)
while the activations rotate through a few buffers.  usage: touch_ab.py [variant codes...]  (tb_gemm_set_variant(2000 + bits): 16 = W, 32 = A, 64 = 128-byte step)"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib()
dev = "cuda"
codes = [int(x) for x in sys.argv[1:]] or [2000, 2016, 2048, 2080, 2112]
def bench(M, N, K, code, reps=5):
    wb = N * K * 2 / 1e6
    nW = min(200, max(4, int(700 / wb) + 1))
    As = [torch.randn(M, K, device=dev).half() for _ in range(3)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nW)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(3)]
    lib.tb_gemm_set_variant(code)
    def run():
        for i in range(nW): ops.gemm(As[i % 3], Ws[i], Os[i % 3])
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    lib.tb_gemm_set_variant(2000)
    return s.elapsed_time(e) / (reps * nW) * 1e3
shapes = [(1848, 2304, 768), (1848, 768, 768), (1848, 3072, 768), (1848, 768, 3072), (1232, 768, 3072), (1232, 3072, 768),
          (2048, 1280, 1280), (2048, 1280, 5120), (2048, 1280, 2560), (512, 1280, 1280), (512, 1280, 2560), (2048, 3840, 1280), (8192, 640, 640)]
print("shape".ljust(24) + "".join(f"{c:>9d}" for c in codes))
for M, N, K in shapes:
    r = [bench(M, N, K, c) for c in codes]
    cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
    print(f"{M}x{N}x{K}".ljust(24) + "".join(f"{x:9.1f}" for x in r) + f"   tile {cfg[0]}x{cfg[1]} kt{cfg[3]} S{cfg[4]}", flush=True)
