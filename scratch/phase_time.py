import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for B, C, H in ((8, 320, 64), (8, 640, 32), (8, 1280, 16)):
    Ho = H // 2
    dy = torch.randn(B * Ho * Ho, C, device="cuda").half()
    wd = (torch.randn(C, 9 * C, device="cuda") / 50).half()
    dx = torch.empty(B * H * H, C, device="cuda", dtype=torch.float16)
    geo = dict(B=B, Hin=Ho, Win=Ho, Cin=C, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=1)
    for phase in (0, 1):
        L.lib().tb_gemm_set_variant(9900 + phase)
        print(f"C={C} {H}x{H}: phase={phase}: {timeit(lambda: ops.gemm(dy, wd, dx, conv=geo)):.1f} us", flush=True)
L.lib().tb_gemm_set_variant(9901)
