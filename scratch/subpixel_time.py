"""upsampler convolutions: materialised nearest-x2 + 9-tap conv (+ dgrad + 2x2 pooling) vs the sub-pixel forms, graph-replayed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def bench(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * 4) * 1e3
for B, C, Hc in [(8, 640, 32), (8, 1280, 16), (8, 1280, 8)]:
    Hf = 2 * Hc
    x = torch.randn(B * Hc * Hc, C, device=dev).half(); w = (torch.randn(C, C, 3, 3, device=dev) / (3 * C ** 0.5)).half(); bias = torch.randn(C, device=dev)
    w9 = w.permute(0, 2, 3, 1).reshape(C, -1).contiguous(); w9d = w.permute(1, 2, 3, 0).reshape(C, -1).contiguous()
    xu = torch.empty(B * Hf * Hf, C, device=dev, dtype=torch.float16); out = torch.empty_like(xu)
    dy = torch.randn(B * Hf * Hf, C, device=dev).half(); du = torch.empty_like(xu); dx = torch.empty_like(x)
    g9 = dict(B=B, Hin=Hf, Win=Hf, Cin=C, Hout=Hf, Wout=Hf, stride=1, sign=1, upsample=0, transposed=0)
    def f9():
        ops.upsample2x(x, xu, B, Hc, Hc, C); ops.gemm(xu, w9, out, bias=bias, conv=g9)
    def b9():
        ops.gemm(dy, w9d, du, conv=dict(g9, sign=-1)); ops.pool2x2_sum(du, dx, B, Hc, Hc, C)
    t9f, t9b = bench(f9), bench(b9)
    if ops.subpixel_ok(B, Hc, Hc, C, C):
        wf, wd = ops.pack_subpixel_weights(w)
        def fs(): ops.gemm(x, wf, out, bias=bias, conv=dict(B=B, Hin=Hc, Win=Hc, Cin=C, Hout=Hf, Wout=Hf, stride=1, sign=1, upsample=2, transposed=0))
        def bs(): ops.gemm(dy, wd, dx, conv=dict(B=B, Hin=Hf, Win=Hf, Cin=C, Hout=Hc, Wout=Hc, stride=1, sign=1, upsample=3, transposed=0))
        tsf, tsb = bench(fs), bench(bs)
    else:
        tsf = tsb = float("nan")
    gf = 2.0 * B * Hf * Hf * C * 9 * C / 1e6
    print(f"B={B} C={C} {Hc}->{Hf}: forward 9-tap {t9f:6.1f} us ({gf/t9f:5.0f} TF/s)  sub-pixel {tsf:6.1f} us ({gf/2.25/tsf:5.0f} TF/s)   dgrad 9-tap {t9b:6.1f}  sub-pixel {tsb:6.1f} us", flush=True)
