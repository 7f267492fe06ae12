"""conv (split-K) + reducer + GroupNorm vs conv + GroupNorm-off-the-slices, graph-replayed, per shape (round 4)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
def bench(fn, reps=20):
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
for B, Cin, Cout, H, res in [(8, 1280, 1280, 16, True), (8, 2560, 1280, 16, False), (8, 1280, 1280, 8, True), (8, 2560, 1280, 8, False), (8, 1920, 1280, 16, True)]:
    M, HW = B * H * H, H * H
    x = torch.randn(M, Cin, device=dev).half(); w = (torch.randn(Cout, 9 * Cin, device=dev) / 100).half()
    bias = torch.randn(Cout, device=dev); rb = torch.randn(B, Cout, device=dev); R = torch.randn(M, Cout, device=dev).half() if res else None
    gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    ws = torch.empty(ops.groupnorm_ws(B, HW, Cout), device=dev); st = torch.empty(B, 32, 2, device=dev)
    h, y = torch.empty(M, Cout, device=dev, dtype=torch.float16), torch.empty(M, Cout, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    def three():
        ops.gemm(x, w, h, conv=geo, bias=bias, rowbias=rb, rows_per_group=HW, R=R)
        ops.groupnorm_fwd(h, y, gamma, beta, st, ws, B, HW, Cout, silu=True)
    def two():
        pk = ops.gemm(x, w, h, conv=geo, bias=bias, rowbias=rb, rows_per_group=HW, R=R, defer=True)
        ops.groupnorm_fwd(h, y, gamma, beta, st, ws, B, HW, Cout, silu=True, partials=pk if isinstance(pk, ops.SplitKPartials) else None)
    def conv_only():
        ops.gemm(x, w, h, conv=geo, bias=bias, rowbias=rb, rows_per_group=HW, R=R, defer=True)
    pk = ops.gemm(x, w, h, conv=geo, defer=True)
    S = pk.S if isinstance(pk, ops.SplitKPartials) else 1
    print(f"B={B} {Cin}->{Cout} @{H}x{H} S={S}: conv+reduce+GN {bench(three):6.1f} us   conv+GN(splitk) {bench(two):6.1f} us   conv alone {bench(conv_only):6.1f} us", flush=True)
