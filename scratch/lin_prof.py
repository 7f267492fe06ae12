"""plain Linear shapes on the wide-tile kernel: back-to-back time, for store / MFMA ablation builds (TB_LIB_SUFFIX)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def b2b(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
res = []
for M, N, K, ld in [(32768, 320, 320, 320), (32768, 960, 320, 960), (32768, 320, 320, 960), (32768, 2560, 320, 2560), (8192, 1920, 640, 1920), (8192, 640, 640, 640)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half()
    buf = torch.empty(M, ld, device=dev, dtype=torch.float16); out = buf[:, :N]
    t = b2b(lambda: ops.gemm(A, W, out))
    byts = 2 * (M * K + N * K + M * N)
    res.append(f"{M}x{N}x{K} ldc={ld}: {t:6.1f} us {byts / t / 1e6:5.2f} TB/s")
print(os.environ.get("TB_LIB_SUFFIX", "base"), " | ".join(res))
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
for M, N, K in [(32768, 320, 320), (32768, 960, 320), (32768, 320, 1280)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    for _ in range(3): ops.gemm(A, W, out)
    L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(A, W, out); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
    d = dbg.tolist()
    print(f"  {M}x{N}x{K} first wg clocks: prologue {d[1]-d[0]} loop {d[2]-d[1]} ({(d[2]-d[1])//(K//64)} per k-step) staging {d[5]-d[2]} pass0 stores {d[3]-d[5]} pass1 {d[4]-d[3]} total {d[4]-d[0]}")
