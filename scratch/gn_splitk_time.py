"""graph-replayed timing of the split-K GroupNorm launches (forward / backward) at the 16x16 and 8x8 map shapes"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib(); dev = "cuda"
def graph_time(fn, n=20, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3
for HW, C, S in [(256, 1280, 4), (256, 2560, 4), (256, 1920, 4), (64, 1280, 10), (64, 2560, 10)]:
    B = 8; M = B * HW; npad = C
    part = torch.randn(S, M, npad, device=dev); bias = torch.randn(C, device=dev); rb = torch.randn(B, C, device=dev)
    R = torch.randn(M, C, device=dev).half()
    x = torch.empty(M, C, device=dev, dtype=torch.half); y = torch.empty_like(x); dx = torch.empty_like(x); add = torch.randn_like(x)
    ga = torch.ones(C, device=dev); be = torch.zeros(C, device=dev); st = torch.empty(B, 32, 2, device=dev)
    f = lambda: L.check(lib.tb_groupnorm_fwd_splitk(L.ptr(part), S, npad, L.ptr(bias), L.ptr(rb), C, L.ptr(R), C, L.ptr(x), C, L.ptr(y), C, L.ptr(ga), L.ptr(be),
                                                    L.ptr(st), B, HW, C, 32, 1e-5, 1, L.stream()), "f")
    b = lambda: L.check(lib.tb_groupnorm_bwd_splitk(L.ptr(part), S, npad, L.ptr(x), C, L.ptr(ga), L.ptr(be), L.ptr(st), L.ptr(add), C, L.ptr(dx), C, B, HW, C, 32,
                                                    1, L.stream()), "b")
    tf, tb = graph_time(f), graph_time(b)
    bf, bb = (4 * S + 6) * M * C, (4 * S + 6) * M * C
    print(f"HW={HW:4d} C={C:5d} S={S:2d}: fwd {tf:6.1f} us ({bf/tf/1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({bb/tb/1e6:5.2f} TB/s)")
