"""8x8-map conv (M = 512, 1280 -> 1280, the 22 launches of the lowest level): stage depth (tb_gemm_set_variant 0 / 4 / 5) x split-K target"""
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
for (B, H, Ci, Co) in [(8, 8, 1280, 1280), (8, 8, 2560, 1280), (8, 16, 1280, 1280)]:
    x = torch.randn(B * H * H, Ci, device="cuda").half(); w = (torch.randn(Co, 9 * Ci, device="cuda") / 100).half()
    out = torch.empty(B * H * H, Co, device="cuda", dtype=torch.float16); bias = torch.randn(Co, device="cuda")
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    fn = lambda: ops.gemm(x, w, out, bias=bias, conv=geo)
    row = []
    for var in (0, 4, 5):
        for tgt in (128, 256, 384, 512, 768):
            lib.tb_gemm_set_variant(var); lib.tb_gemm_set_variant(1000 + tgt)
            t = b2b(fn)
            import ctypes
            cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
            row.append(f"v{var}/t{tgt}: {t:5.1f} (S{cfg[4]},{cfg[0]}x{cfg[1]})")
    lib.tb_gemm_set_variant(0); lib.tb_gemm_set_variant(1384)
    print(f"{Ci}->{Co} @{H}x{H}: " + " | ".join(row))
