"""A/B of the whole captured step under alternating ATTRIBUTES of the step object (sustained, interleaved; same method as scratch/ab_step.py).
usage: ab_attr.py name:attr=value[,attr=value..] ...   (value: int / float / True / False; attr may be dotted: unet.foo)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2): step.step_eager()
def setattr_dotted(obj, path, val):
    if path.startswith("unetmod."):   # a module global of textboost_amd.unet (read at issue time), e.g. unetmod.FF_CHAIN=0
        import textboost_amd.unet as U
        setattr(U, path.split(".", 1)[1], val)
        return
    parts = path.split(".")
    for p in parts[:-1]: obj = getattr(obj, p)
    setattr(obj, parts[-1], val)
def parse(v):
    if v in ("True", "False"): return v == "True"
    try: return int(v)
    except ValueError: return float(v)
configs = []
for a in sys.argv[1:]:
    name, kv = a.split(":", 1)
    configs.append((name, [(x.split("=")[0], parse(x.split("=")[1])) for x in kv.split(",") if x]))
graphs = {}
for name, kvs in configs:
    for k, v in kvs: setattr_dotted(step, k, v)
    step.step_eager(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step.draw(); step.forward_backward(); step.optimizer_step()
    graphs[name] = g
res = {n: [] for n, _ in configs}
for rnd in range(3):
    for name, _ in configs:
        g = graphs[name]
        for _ in range(80): g.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100): g.replay()
        e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 100)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:24s} median {v[len(v)//2]:7.3f} ms  min {v[0]:7.3f}  ({1000/v[len(v)//2]:.2f} steps/s)")
