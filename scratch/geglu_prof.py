"""GEGLU projection GEMM (ff.net.0.proj) phase times: back-to-back launch time with / without the C2 store, against a plain Linear of the same
shape, and s_memtime stamps of the first / last workgroup (start, prologue done, main loop done, epilogue done)."""
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
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
for M, C in [(32768, 320), (8192, 640), (2048, 1280)]:
    N = 8 * C
    A = torch.randn(M, C, device=dev).half(); W = (torch.randn(N, C, device=dev) / C ** 0.5).half(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N // 2, device=dev, dtype=torch.float16); raw = torch.empty(M, N, device=dev, dtype=torch.float16)
    full = torch.empty(M, N, device=dev, dtype=torch.float16)
    t_full = b2b(lambda: ops.gemm(A, W, out, bias=bias, act=L.ACT_GEGLU, C2=raw))
    t_noc2 = b2b(lambda: ops.gemm(A, W, out, bias=bias, act=L.ACT_GEGLU))
    t_lin = b2b(lambda: ops.gemm(A, W, full, bias=bias))
    L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(A, W, out, bias=bias, act=L.ACT_GEGLU, C2=raw); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
    d = dbg.tolist()
    print(f"M={M} C={C}: geglu {t_full:.1f} us, without C2 {t_noc2:.1f}, plain Linear [M,{N}] {t_lin:.1f} | first wg: prologue {d[1]-d[0]} loop {d[2]-d[1]} epilogue {d[3]-d[2]} (post-loop barrier {d[6]-d[2]} bias {d[7]-d[6]} staging {d[4]-d[7]} barrier {d[5]-d[4]} units {d[3]-d[5]})"
          f" | last wg: start +{d[8]-d[0]} prologue {d[9]-d[8]} loop {d[10]-d[9]} epilogue {d[11]-d[10]} end +{d[11]-d[0]}")
