"""8x8-map convolutions: 5-slot weight ring, one workgroup per CU (tb_gemm_set_variant 7007, round 5) against the 2-slot halo kernel (7003); cold weights"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
NR = 6
def timeit(fn, reps=12):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NR)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
B, H = 8, 8
for Ci, Co, sign in ((1280, 1280, 1), (2560, 1280, 1), (1280, 2560, -1), (1280, 1280, -1)):
    x = [torch.randn(B * H * H, Ci, device="cuda").half() for _ in range(NR)]
    w = [(torch.randn(Co, 9 * Ci, device="cuda") / (9 * Ci) ** 0.5).half() for _ in range(NR)]
    out = [torch.empty(B * H * H, Co, device="cuda", dtype=torch.float16) for _ in range(NR)]
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=sign, upsample=0, transposed=0)
    res = []
    for v in ((7011, 7003, 7011, 7003) if os.environ.get('V') is None else (int(os.environ['V']),) * 4):
        L.lib().tb_gemm_set_variant(v)
        import ctypes
        res.append(timeit(lambda i: ops.gemm(x[i], w[i], out[i], conv=geo)))
    L.lib().tb_gemm_set_variant(7011)
    print(f"{Ci}->{Co} sign {sign}: 12-pitch permuted {res[0]:.1f} / {res[2]:.1f} us, round-4 layout {res[1]:.1f} / {res[3]:.1f} us (incl. reducer)")
