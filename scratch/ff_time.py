"""Fused GEGLU feed-forward (tb_ff_fwd / tb_ff_bwd) against the launches it replaces, M = 32768 (the 64x64 maps at B = 8), graph-replayed, rotating buffers."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_gemm import pack_geglu
M, C, I = 32768, 320, 1280
NB = 4
torch.manual_seed(0)
w1 = (torch.randn(2 * I, C, device="cuda") / C ** 0.5).half(); b1 = torch.randn(2 * I, device="cuda") * 0.3
w2 = (torch.randn(C, I, device="cuda") / I ** 0.5).half(); b2 = torch.randn(C, device="cuda") * 0.3
w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
w2d, w1d = w2.t().contiguous(), w1p.t().contiguous()
xs = [torch.randn(M, C, device="cuda").half() for _ in range(NB)]
Rs = [torch.randn(M, C, device="cuda").half() for _ in range(NB)]
hgs = [torch.empty(M, 2 * I, device="cuda", dtype=torch.float16) for _ in range(NB)]
ys = [torch.empty(M, C, device="cuda", dtype=torch.float16) for _ in range(NB)]
gated = torch.empty(M, I, device="cuda", dtype=torch.float16)
dproj = torch.empty(M, 2 * I, device="cuda", dtype=torch.float16)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * NB) * 1e3
def fused_f():
    for i in range(NB): ops.ff_fwd(xs[i], w1p, b1p, w2, b2, hgs[i], ys[i], R=Rs[i])
def two_f():
    for i in range(NB):
        ops.gemm(xs[i], w1p, gated, bias=b1p, act=L.ACT_GEGLU, C2=hgs[i]); ops.gemm(gated, w2, ys[i], bias=b2, R=Rs[i])
def fused_b():
    for i in range(NB): ops.ff_bwd(xs[i], w2d, w1d, hgs[i], ys[i])
def two_b():
    for i in range(NB):
        ops.gemm(xs[i], w2d, dproj, act=L.ACT_GEGLU_GRAD, C2=hgs[i]); ops.gemm(dproj, w1d, ys[i])
two_f()
for name, f in [("fwd two launches", two_f), ("fwd fused", fused_f), ("bwd two launches", two_b), ("bwd fused", fused_b)]:
    t = timeit(f)
    print(f"{name:20s} {t:8.1f} us   {2.0 * M * C * 3 * I / t / 1e6:7.1f} TFLOP/s", flush=True)
if os.environ.get("TB_LIB_SUFFIX"):   # FF_PROF build: per-phase cycle sums of waves 0 / 4 of workgroup 0
    dbg = torch.zeros(16, device="cuda", dtype=torch.int64)
    L.lib().tb_ff_debug(L.ptr(dbg))
    for name, f in [("fwd", lambda: ops.ff_fwd(xs[0], w1p, b1p, w2, b2, hgs[0], ys[0], R=Rs[0])), ("bwd", lambda: ops.ff_bwd(xs[0], w2d, w1d, hgs[0], ys[0]))]:
        f(); torch.cuda.synchronize(); dbg.zero_(); f(); torch.cuda.synchronize()
        d = dbg.tolist()
        for w in (0, 1):
            v = d[8 * w: 8 * w + 6]
            print(f"{name} wave {4 * w}: top(wait+issue) {v[0]/40:7.0f}  phaseA {v[1]/40:7.0f}  epiA {v[2]/40:7.0f}  barrier {v[3]/40:7.0f}  phaseB {v[4]/40:7.0f}  cycles per tile; loop total {v[5]}")
