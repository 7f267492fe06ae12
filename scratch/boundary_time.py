"""conv_in / conv_out / conv_out-dgrad at the metric's shape (B = 8, 64 x 64, 320 channels): VALU kernels vs the MFMA form, graph-replayed"""
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
B, H, W, C = 8, 64, 64, 320
x = torch.randn(B, 4, H, W, device=dev).half(); wp = torch.randn(36, C, device=dev) * 0.1; b = torch.randn(C, device=dev)
out = torch.empty(B * H * W, C, device=dev, dtype=torch.float16)
h = torch.randn(B * H * W, C, device=dev).half(); wop = torch.randn(4, 9, C, device=dev) * 0.1; bo = torch.randn(4, device=dev)
pred = torch.empty(B, 4, H, W, device=dev, dtype=torch.float16)
dpred = torch.randn(B, 4, H, W, device=dev); dh = torch.empty_like(out)
for v in (0, 1):
    lib.tb_boundary_conv_set_variant(v)
    t1 = graph_time(lambda: ops.conv4_to_nhwc(x, wp, b, out, B, H, W, C, sign=1))
    t2 = graph_time(lambda: ops.conv_to4(h, wop, bo, pred, B, H, W, C))
    t3 = graph_time(lambda: ops.conv4_to_nhwc(dpred, wp, None, dh, B, H, W, C, sign=-1))
    print(f"variant {v}: conv_in {t1:6.1f} us  conv_out {t2:6.1f} us  conv_out dgrad {t3:6.1f} us")
