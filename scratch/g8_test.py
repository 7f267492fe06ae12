"""gemm8 (8-wave wide tiles) vs torch fp32 and vs the 4-wave kernels: correctness + A/B timing."""
import sys, os, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
lib = L.lib()
torch.manual_seed(0)
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
def used8():
    c = (ctypes.c_int * 6)()
    return lib.tb_gemm8_last(c), list(c)
def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()
dbg = torch.zeros(16, dtype=torch.int64, device=dev)
def stamps(fn):
    lib.tb_gemm8_debug(L.ptr(dbg)); fn(); fn(); torch.cuda.synchronize(); lib.tb_gemm8_debug(None)
    d = dbg.tolist()
    f = lambda o: " ".join(f"{(d[o+k]-d[o])/2100.0:6.2f}" for k in (1, 2, 5, 6, 3, 4))   # s_memtime ~ shader clock, ~2.1 GHz
    return f"first[{f(0)}] last[{f(8)}] us(start->prologue,loop,epi0 staged,epi0 units done,epi0 barrier,epi1)"
print("== linear")
for M, N, K, extra in [(32768, 320, 320, "R"), (32768, 960, 320, ""), (32768, 320, 1280, "b"), (32768, 1280, 320, ""), (8192, 640, 640, "Rb"), (8192, 1920, 640, ""), (8192, 640, 2560, ""), (256 * 128, 320, 64, "")]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    kw = {}
    if "R" in extra: kw["R"] = torch.randn(M, N, device=dev).half()
    if "b" in extra: kw["bias"] = torch.randn(N, device=dev)
    lib.tb_gemm8_set(3); ops.gemm(A, W, out, **kw); u, cfg = used8()
    ref = A.float() @ W.float().t()
    if "R" in kw: ref += kw["R"].float()
    if "bias" in kw: ref += kw["bias"]
    e8 = rel(out, ref); mx = (out.float() - ref).abs().max().item()
    t8 = timeit(lambda: ops.gemm(A, W, out, **kw))
    print("   ", stamps(lambda: ops.gemm(A, W, out, **kw)))
    lib.tb_gemm8_set(0); out2 = torch.empty_like(out); ops.gemm(A, W, out2, **kw); e4 = rel(out2, ref)
    t4 = timeit(lambda: ops.gemm(A, W, out2, **kw))
    print(f"  {M:6d} {N:5d} {K:5d} {extra:3s} used8={u} {cfg}  err8 {e8:.2e} (max {mx:.3f}) err4 {e4:.2e}  t8 {t8*1e6:7.1f} us ({2*M*N*K/t8/1e12:6.1f} TF)  t4 {t4*1e6:7.1f} us ({2*M*N*K/t4/1e12:6.1f} TF)")
print("== conv3x3")
import torch.nn.functional as F
for Ci, Co, H, dg in [(320, 320, 64, 0), (320, 320, 64, 1), (640, 320, 64, 0), (960, 320, 64, 0), (320, 640, 64, 0), (640, 640, 32, 0), (640, 640, 32, 1), (1280, 640, 32, 0), (640, 1280, 32, 0), (64, 320, 64, 0), (64, 160, 64, 0)]:
    B = 8
    x = torch.randn(B, Ci, H, H, device=dev).half()
    w = (torch.randn(Co, Ci, 3, 3, device=dev) / (9 * Ci) ** 0.5).half()
    bias = torch.randn(Co, device=dev)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * H, Ci).contiguous()
    if dg:   # dgrad form: weights [Co(out of this op), 9*Ci] with flipped taps (sign -1)
        wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
        ref = F.conv2d(x.float(), w.float().flip(2, 3), padding=1)
    else:
        wp = w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
        ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * H, Co)
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=-1 if dg else 1, upsample=0, transposed=0)
    out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
    kw = {} if dg else {"bias": bias}
    lib.tb_gemm8_set(3); ops.gemm(xn, wp, out, conv=geo, **kw); u, cfg = used8()
    e8 = rel(out, ref); mx = (out.float() - ref).abs().max().item()
    t8 = timeit(lambda: ops.gemm(xn, wp, out, conv=geo, **kw))
    print("   ", stamps(lambda: ops.gemm(xn, wp, out, conv=geo, **kw)))
    lib.tb_gemm8_set(0); out2 = torch.empty_like(out); ops.gemm(xn, wp, out2, conv=geo, **kw); e4 = rel(out2, ref)
    t4 = timeit(lambda: ops.gemm(xn, wp, out2, conv=geo, **kw))
    fl = 2 * B * H * H * Co * 9 * Ci
    print(f"  {Ci:5d}->{Co:5d} @{H:3d} dg={dg} used8={u} {cfg}  err8 {e8:.2e} (max {mx:.3f}) err4 {e4:.2e}  t8 {t8*1e6:7.1f} us ({fl/t8/1e12:6.1f} TF)  t4 {t4*1e6:7.1f} us ({fl/t4/1e12:6.1f} TF)")
lib.tb_gemm8_set(3)
