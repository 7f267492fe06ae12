"""s_memtime stamps (tb_gemm8_debug) of the one-per-CU Linear tiles: where the fixed cost of a K = 640 launch goes (cold operands: rotating buffers)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
NR = int(os.environ.get("NR", "8"))
for M, N, K, res in [(8192, 640, 640, True), (8192, 640, 640, False), (2048, 1280, 1280, True), (8192, 640, 2560, True)]:
    A = [torch.randn(M, K, device=dev).half() for _ in range(NR)]
    W = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(NR)]
    R = [torch.randn(M, N, device=dev).half() for _ in range(NR)]
    out = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(NR)]
    b = torch.randn(N, device=dev)
    for i in range(NR): ops.gemm(A[i], W[i], out[i], bias=b, R=R[i] if res else None)
    rows = []
    for i in range(NR):
        L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(A[i], W[i], out[i], bias=b, R=R[i] if res else None); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
        d = dbg.tolist()
        rows.append((d[1]-d[0], d[2]-d[1], d[6]-d[2], d[5]-d[6], d[3]-d[5], d[3]-d[0], d[8+1]-d[8], d[8+2]-d[8+1], d[8+3]-d[8+2], d[8+3]-d[8]))
    r = [sorted(x)[len(x)//2] for x in zip(*rows)]
    print(f"{M}x{N}x{K} res={res}: first wg: prologue {r[0]} loop {r[1]} ({r[1]//(K//64)}/step) drain {r[2]} staging {r[3]} units+stores {r[4]} total {r[5]} | last wg: prologue {r[6]} loop {r[7]} epi {r[8]} total {r[9]}  (cycles of the 100 MHz?/shader clock)")
if os.environ.get("PSTAMP"):
    for M, N, K in [(8192, 640, 640), (2048, 1280, 1280)]:
        A = [torch.randn(M, K, device=dev).half() for _ in range(NR)]
        W = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(NR)]
        out = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(NR)]
        rows = []
        for i in range(NR):
            L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(A[i], W[i], out[i]); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
            d = dbg.tolist()
            rows.append((d[7]-d[0], d[3]-d[7], d[4]-d[3], d[5]-d[4], d[1]-d[5]))
        r = [sorted(x)[len(x)//2] for x in zip(*rows)]
        print(f"{M}x{N}x{K} prologue: entry->tile math/bias/rs {r[0]}, ->operand addressing {r[1]}, ->stages issued {r[2]}, ->wait done {r[3]}, ->barrier {r[4]}")
