"""How much of the sustained step is power: replay the SAME step graph (data-independent instruction streams) with every frozen weight zeroed.
On zeros the matrix pipe draws little and the board keeps its clock; the difference to the real-data replay is what the power limit costs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step.draw(); step.forward_backward(); step.optimizer_step()
def timeit():
    for _ in range(60): g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 100
print(f"real weights : {timeit():.3f} ms per step", flush=True)
def tensors_of(obj, seen, depth=0):
    if id(obj) in seen or depth > 4: return
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.is_floating_point(): yield obj
    elif isinstance(obj, dict):
        for v in obj.values(): yield from tensors_of(v, seen, depth + 1)
    elif isinstance(obj, (list, tuple)):
        for v in obj: yield from tensors_of(v, seen, depth + 1)
    elif hasattr(obj, "__dict__") and type(obj).__module__.startswith("textboost_amd"):
        for v in vars(obj).values(): yield from tensors_of(v, seen, depth + 1)
n = 0; by = 0
for t in tensors_of(step, set()):
    t.zero_(); n += 1; by += t.numel() * t.element_size()
torch.cuda.synchronize()
print(f"zeroed {n} tensors, {by / 1e9:.2f} GB", flush=True)
print(f"all zeros    : {timeit():.3f} ms per step", flush=True)
