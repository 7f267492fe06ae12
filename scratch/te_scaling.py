"""How does the text encoder's forward / backward time scale with the number of rows (latency-bound launches)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2):
    step.step_eager()
te, B = step.te, step.B

def timeit(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        g.replay()
    e.record(); torch.cuda.synchronize()
    print(f"{name:44s} {s.elapsed_time(e)/n:8.3f} ms", flush=True)

ids16 = step.ids_all
ids8 = step.ids_all[:8].contiguous()
timeit("fwd 8 seq (M=616), slot 1", lambda: te.forward(ids8, slot=1))
timeit("fwd 16 seq (M=1232), slot 1", lambda: te.forward(ids16, slot=1))
timeit("fwd 16 + 8 frozen (M=1848), slot 0", lambda: te.forward(ids16, slot=0, extra_ids=step.prior_ids, extra_table=step.teacher_table32))
d8 = step.d_all[:616].contiguous()
def bwd8():
    step.flat_grad.zero_(); te.backward(d8, slot=1)
te.forward(ids8, slot=1)
timeit("bwd 8 seq (M=616)", bwd8)
te.forward(ids16, slot=1)
def bwd16():
    step.flat_grad.zero_(); te.backward(step.d_all, slot=1)
timeit("bwd 16 seq (M=1232)", bwd16)
