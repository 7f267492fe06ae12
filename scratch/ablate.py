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
cases = {}
B, H, Ci, Co = 8, 64, 320, 320
x = torch.randn(B * H * H, Ci, device=dev).half(); w = torch.randn(Co, 9 * Ci, device=dev).half(); out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
cases["conv 320->320@64 (128x64)"] = lambda: ops.gemm(x, w, out, conv=geo)
x2 = torch.randn(8 * 32 * 32, 640, device=dev).half(); w2 = torch.randn(640, 9 * 640, device=dev).half(); out2 = torch.empty(8 * 32 * 32, 640, device=dev, dtype=torch.float16)
geo2 = dict(B=8, Hin=32, Win=32, Cin=640, Hout=32, Wout=32, stride=1, sign=1, upsample=0, transposed=0)
cases["conv 640->640@32 (128x128)"] = lambda: ops.gemm(x2, w2, out2, conv=geo2)
A = torch.randn(2048, 1280, device=dev).half(); W = torch.randn(10240, 1280, device=dev).half(); o3 = torch.empty(2048, 10240, device=dev, dtype=torch.float16)
cases["lin 2048x10240x1280"] = lambda: ops.gemm(A, W, o3)
A4 = torch.randn(32768, 2560, device=dev).half(); W4 = torch.randn(320, 2560, device=dev).half(); o4 = torch.empty(32768, 320, device=dev, dtype=torch.float16)
cases["lin 32768x320x2560"] = lambda: ops.gemm(A4, W4, o4)
for name, fn in cases.items():
    r = []
    for abl in (0, 1, 2, 3, 7):
        L.lib().tb_gemm_set_variant(2000 + abl)
        r.append(timeit(fn))
    L.lib().tb_gemm_set_variant(2000)
    print(f"{name:32s} full {r[0]:7.1f} us | no-loads {r[1]:7.1f} | no-mfma {r[2]:7.1f} | neither {r[3]:7.1f} | +no-epilogue {r[4]:7.1f}")
