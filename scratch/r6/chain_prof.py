import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textboost_amd import ops, _lib as L
M, C = 32768, 320
dev = "cuda"
for N2 in (320, 960):
    x = torch.randn(M, C, device=dev).half(); R = torch.randn(M, C, device=dev).half(); t = torch.empty(M, C, device=dev, dtype=torch.float16)
    w1 = (torch.randn(C, C, device=dev) / 18).half(); b1 = torch.randn(C, device=dev); w2 = (torch.randn(N2, C, device=dev) / 18).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev); st = torch.empty(M, 2, device=dev); y = torch.empty(M, N2, device=dev, dtype=torch.float16)
    dbg = torch.zeros(32, dtype=torch.int64, device=dev)
    L.lib().tb_chain320_debug.argtypes = [ctypes.c_void_p]; L.lib().tb_chain320_debug(ctypes.c_void_p(dbg.data_ptr()))
    for _ in range(3): ops.chain320(x, w1, b1, R, t, g, b, st, w2, None, y)
    torch.cuda.synchronize()
    d = dbg.tolist()
    names = ["start", "loads issued", "tile0 done", "stage1 done", "barrier+issue2", "LN done", "operand", "s2 tile0", "s2 last tile", "copy-out issued", "end"]
    for w in (0, 16):
        base = d[w]
        print(f"N2={N2} wave {w // 4}: " + "  ".join(f"{n}={d[w + i] - base}" for i, n in enumerate(names)))
