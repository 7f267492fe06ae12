"""GroupNorm fwd / bwd per shape: two-pass kernels (variant 1: small-map fused only) vs the one-pass slice kernels (variant 3), graph-replayed."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(6): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * 6) * 1e3
for B, HW, C in [(8, 4096, 320), (8, 4096, 640), (8, 4096, 960), (8, 1024, 640), (8, 1024, 1280), (8, 1024, 1920), (8, 1024, 960), (8, 256, 1920), (8, 256, 640)]:
    M = B * HW
    nb = 3   # rotate buffers so the L2 / MALL does not hold everything
    xs = [torch.randn(M, C, device=dev).half() for _ in range(nb)]; dys = [torch.randn(M, C, device=dev).half() for _ in range(nb)]
    adds = [torch.randn(M, C, device=dev).half() for _ in range(nb)]
    ys = [torch.empty(M, C, device=dev, dtype=torch.float16) for _ in range(nb)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.empty(ops.groupnorm_ws(B, HW, C), device=dev); st = torch.empty(B, 32, 2, device=dev)
    i = [0]
    def fwd():
        k = i[0] % nb; i[0] += 1
        ops.groupnorm_fwd(xs[k], ys[k], gamma, beta, st, ws, B, HW, C, silu=True)
    def bwd():
        k = i[0] % nb; i[0] += 1
        ops.groupnorm_bwd(dys[k], xs[k], gamma, beta, st, ys[k], ws, B, HW, C, silu=True, add=adds[k])
    r = []
    for var in (1, 3):
        L.lib().tb_groupnorm_set_variant(var)
        r.append((bench(fwd), bench(bwd)))
    L.lib().tb_groupnorm_set_variant(11)
    gb = M * C * 2 / 1e3
    print(f"B={B} HW={HW:5d} C={C:5d}: fwd two-pass {r[0][0]:6.1f} us  one-pass {r[1][0]:6.1f} us ({2*gb/r[1][0]/1e3:5.2f} TB/s)   bwd two-pass {r[0][1]:6.1f} us  one-pass {r[1][1]:6.1f} us ({4*gb/r[1][1]/1e3:5.2f} TB/s)", flush=True)
