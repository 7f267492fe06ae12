import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
def check(v):
    L.lib().tb_gemm_set_variant(v)
    torch.manual_seed(0)
    A = torch.randn(300, 640, device=dev).half(); W = (torch.randn(320, 640, device=dev) / 25).half(); o = torch.empty(300, 320, device=dev, dtype=torch.float16)
    ops.gemm(A, W, o); ref = A.float() @ W.float().T
    e1 = ((o.float() - ref).norm() / ref.norm()).item()
    x = torch.randn(2 * 16 * 16, 128, device=dev).half(); w = (torch.randn(192, 9 * 128, device=dev) / 34).half(); out = torch.empty(512, 192, device=dev, dtype=torch.float16)
    geo = dict(B=2, Hin=16, Win=16, Cin=128, Hout=16, Wout=16, stride=1, sign=1, upsample=0, transposed=0)
    ops.gemm(x, w, out, conv=geo)
    xr = x.float().view(2, 16, 16, 128).permute(0, 3, 1, 2); wr = w.float().view(192, 3, 3, 128).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, wr, padding=1).permute(0, 2, 3, 1).reshape(512, 192)
    e2 = ((out.float() - ref).norm() / ref.norm()).item()
    return e1, e2
for v in (0, 1, 2):
    print("variant", v, "rel err lin/conv", check(v))
shapes = [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 5120, 640), (2048, 10240, 1280), (2048, 1280, 1280), (1232, 2304, 832), (1232, 768, 3072), (616, 768, 3072)]
convs = [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8), (2560, 1280, 8), (960, 320, 64), (1920, 640, 32)]
print("linear TF/s by variant (0: BK64x2, 1: BK32x3, 2: BK32x2)")
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    r = []
    for v in (0, 1, 2):
        L.lib().tb_gemm_set_variant(v)
        t = timeit(lambda: ops.gemm(A, W, out)); r.append(2 * M * N * K / t / 1e12)
    print(f"  {M:6d} {N:6d} {K:6d}: " + "  ".join(f"{x:7.1f}" for x in r))
print("conv TF/s by variant")
for Ci, Co, H in convs:
    B = 8
    x = torch.randn(B * H * H, Ci, device=dev).half(); w = torch.randn(Co, 9 * Ci, device=dev).half(); out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    r = []
    for v in (0, 1, 2):
        L.lib().tb_gemm_set_variant(v)
        t = timeit(lambda: ops.gemm(x, w, out, conv=geo)); r.append(2 * B * H * H * Co * 9 * Ci / t / 1e12)
    print(f"  {Ci:5d}->{Co:5d} @{H:3d}: " + "  ".join(f"{x:7.1f}" for x in r))
