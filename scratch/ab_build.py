"""A/B of BUILD-time switches inside one process: one TextBoostStep per variant (module attributes set before build_step), both captured as
graphs, sustained interleaved replays (as scratch/ab_step.py).
usage: ab_build.py name:mod.ATTR=val[,mod.ATTR=val..] ...      e.g.  base:unet.FOLD_LN=0 fold:unet.FOLD_LN=1"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
from textboost_amd import ops, _lib as L
lib = L.lib()
configs = []
for a in sys.argv[1:]:
    name, codes = a.split(":", 1)
    configs.append((name, [c for c in codes.split(",") if c]))
graphs, steps = {}, {}
for name, codes in configs:
    for c in codes:
        k, v = c.split("=", 1)
        if k.startswith("env."):
            os.environ[k[4:]] = v; continue
        mod, attr = k.rsplit(".", 1)
        m = importlib.import_module("textboost_amd." + mod)
        setattr(m, attr, type(getattr(m, attr))(int(v)) if isinstance(getattr(m, attr), (bool, int)) else v)
    step, _ = build_step(batch=int(os.environ.get("TB_AB_BATCH", "8")), latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
    for _ in range(2): step.step_eager()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step.draw(); step.forward_backward(); step.optimizer_step()
    graphs[name], steps[name] = g, step
    print(name, "loss", step.scalars().get("loss_mse"), flush=True)
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
