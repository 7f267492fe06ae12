"""time a few gemm8 shapes with the library selected by TB_LIB_SUFFIX (ablation builds: results are garbage, only the time matters)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; B = 8
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
res = []
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
PROF = os.environ.get("G8_PROF") == "1"
for Ci, Co, H in [(64, 320, 64), (320, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32)]:
    x = torch.randn(B * H * H, Ci, device=dev).half(); w = (torch.randn(Co, 9 * Ci, device=dev) / (9 * Ci) ** 0.5).half()
    out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    res.append(f"{Ci}->{Co}@{H}: {timeit(lambda: ops.gemm(x, w, out, conv=geo)):7.1f}")
    if PROF:
        L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(x, w, out, conv=geo); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
        d = dbg.tolist(); nst = 9 * Ci // 64
        print(f"  {Ci}->{Co}@{H} clocks per step: X[load {d[16]/nst:.0f} bar {d[17]/nst:.0f} mfma {d[18]/nst:.0f} bar {d[19]/nst:.0f}] Y[load {d[20]/nst:.0f} bar {d[21]/nst:.0f} mfma {d[22]/nst:.0f} bar {d[23]/nst:.0f}] total {(d[3]-d[0])/1:.0f} loop {(d[2]-d[1]):.0f}")
for M, N, K in [(32768, 320, 320), (32768, 320, 1280), (32768, 1280, 320), (8192, 640, 640), (8192, 640, 2560)]:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    res.append(f"{M}x{N}x{K}: {timeit(lambda: ops.gemm(A, W, out)):7.1f}")
print(os.environ.get("TB_LIB_SUFFIX", "base"), " | ".join(res))
