"""256-pixel x 80-channel convolution tile: 4-slot ring + pieces between the MFMAs (round 5, DMACH) against round 4's 3-slot loop (tb_gemm8_set bit 32768)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
NR = 4
def timeit(fn, reps=12):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn(i % NR)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
B = 8
for Ci, Co, H, sign in ((640, 640, 32, 1), (1280, 640, 32, 1), (320, 640, 32, 1), (640, 640, 32, -1), (640, 1280, 32, -1), (1920, 640, 32, 1)):
    x = [torch.randn(B * H * H, Ci, device="cuda").half() for _ in range(NR)]
    w = [(torch.randn(Co, 9 * Ci, device="cuda") / (9 * Ci) ** 0.5).half() for _ in range(NR)]
    out = [torch.empty(B * H * H, Co, device="cuda", dtype=torch.float16) for _ in range(NR)]
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=sign, upsample=0, transposed=0)
    res = []
    for bits in (39, 39 | 32768, 39, 39 | 32768):
        L.lib().tb_gemm8_set(bits)
        res.append(timeit(lambda i: ops.gemm(x[i], w[i], out[i], conv=geo)))
    L.lib().tb_gemm8_set(39)
    print(f"{Ci}->{Co} @{H} sign {sign}: 4-slot DMACH {res[0]:.1f} / {res[2]:.1f} us, 3-slot {res[1]:.1f} / {res[3]:.1f} us")
