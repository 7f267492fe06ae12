import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("conv us: gather | halo | halo<64> forced  (TF/s halo)")
for Ci, Co, H in [(320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32), (1280, 1280, 16), (2560, 1280, 16), (320, 640, 32), (1280, 1280, 8), (2560, 1280, 8)]:
    B = 8
    x = torch.randn(B * H * H, Ci, device=dev).half(); w = torch.randn(Co, 9 * Ci, device=dev).half(); out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    r = []
    for v, n in ((7000, 9000), (7001, 9000), (7001, 9001)):
        L.lib().tb_gemm_set_variant(v); L.lib().tb_gemm_set_variant(n)
        r.append(timeit(lambda: ops.gemm(x, w, out, conv=geo)))
    L.lib().tb_gemm_set_variant(9000)
    fl = 2 * B * H * H * Co * 9 * Ci
    print(f"  {Ci:5d}->{Co:5d} @{H:3d}: {r[0]:8.1f} | {r[1]:8.1f} | {r[2]:8.1f}  ({fl/r[1]/1e6:6.1f} TF/s)")
