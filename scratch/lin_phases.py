"""G8_PROF build (TB_LIB_SUFFIX=_prof): per-phase s_memtime sums of waves 0 (group 0) and 4 (group 1) of workgroup 0 for the one-per-CU Linear tiles"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
dbg = torch.zeros(32, dtype=torch.int64, device=dev)
for M, N, K in [(8192, 640, 640), (2048, 1280, 1280), (8192, 640, 2560), (8192, 1920, 640)]:
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) / K ** 0.5).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    b = torch.randn(N, device=dev)
    for _ in range(3): ops.gemm(A, W, out, bias=b)
    rows = []
    for _ in range(5):
        L.lib().tb_gemm8_debug(L.ptr(dbg)); ops.gemm(A, W, out, bias=b); torch.cuda.synchronize(); L.lib().tb_gemm8_debug(None)
        rows.append(dbg.tolist())
    d = [sorted(x)[2] for x in zip(*rows)]
    steps = K // 64
    import ctypes
    c8 = (ctypes.c_int * 6)(); L.lib().tb_gemm8_last(c8)
    print(f"{M}x{N}x{K} tile {list(c8)}: per k-step  group0: LOAD {d[16]//steps} bar {d[17]//steps} COMPUTE {d[18]//steps} bar {d[19]//steps} | group1: LOAD {d[20]//steps} bar {d[21]//steps} COMPUTE {d[22]//steps} bar {d[23]//steps} | loop {(d[2]-d[1])//steps}/step, total {d[3]-d[0]}")
