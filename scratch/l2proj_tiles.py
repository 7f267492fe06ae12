"""16x16-map projection (2048 x 1280 x 1280, 50 launches per step) under the 4-wave kernel's tile shapes (tb_gemm_set_variant(8000 + v)),
with and without residual, and the gemm8 64x320 tile (not selected: 128 tiles)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib()
def b2b(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in [(2048, 1280, 1280), (2048, 1280, 3840), (2048, 3840, 1280), (1848, 768, 768), (1848, 3072, 768)]:
    A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16); R = torch.randn(M, N, device="cuda").half(); bias = torch.randn(N, device="cuda")
    row = []
    for v, name in [(0, "auto"), (1, "64x64"), (2, "128x64"), (3, "128x128")]:
        lib.tb_gemm_set_variant(8000 + v)
        t = b2b(lambda: ops.gemm(A, W, out, bias=bias, R=R))
        import ctypes
        cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
        row.append(f"{name}: {t:5.1f} us ({cfg[0]}x{cfg[1]} ns{cfg[3] % 10} S{cfg[4]})")
    lib.tb_gemm_set_variant(8000)
    print(f"{M}x{N}x{K}: " + " | ".join(row))
