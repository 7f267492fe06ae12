"""Time the parts of one step in isolation (eager launches, HIP events): CLIP student fwd / teacher fwd / CLIP bwd / UNet fwd / UNet bwd / optimizer."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
from textboost_amd import ops, _lib as L
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2):
    step.step_eager()
te, B = step.te, step.B
BT = B * te.T

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
    print(f"{name:28s} {s.elapsed_time(e)/n:8.3f} ms (graph replay)")

timeit("clip student fwd (M=1232)", lambda: te.forward(step.ids_all, slot=0))
timeit("clip teacher fwd (M=616)", lambda: step.teacher.forward(step.prior_ids, slot=0))
def bwd():
    step.flat_grad.zero_(); te.backward(step.d_all, slot=0)
timeit("clip bwd (M=1232)", bwd)
timeit("pack_lora", lambda: te.pack_lora())
timeit("unet fwd", lambda: step.unet.forward(step.noisy, step.timesteps, step.ehs16))
timeit("unet bwd", lambda: step.unet.backward(step.dpred, d_ehs_out=step.d_ehs))
timeit("optimizer", lambda: step.optimizer_step())
timeit("draw+add_noise", lambda: (step.draw(), ops.add_noise(step.x0, step.noise, step.timesteps, step.acp, step.noisy, step.velocity)))
