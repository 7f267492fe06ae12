"""round 5: the step as ONE graph against TWO graphs (A = draw + forward, B = backward + optimizer) replayed alternately -- does the submission of a
graph overlap the execution of the other one (the ~0.25 ms bubble ~20 launches into every replay)?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(3): step.step_eager()
torch.cuda.synchronize()
def cap(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
def full():
    step.draw(); step.forward_backward(); step.optimizer_step()
def part_a():
    step.draw(); step._phase_student(); step._phase_unet_forward()
    if step.kpl and step.merge_teacher: step._phase_teacher()
def part_b():
    step._phase_unet_backward(); step._phase_encoder_backward(); step.optimizer_step()
NPARTS = int(os.environ.get("NPARTS", "2"))
g1 = cap(full)
ga, gb = cap(part_a), cap(part_b)
def run_one(n):
    for _ in range(n): g1.replay()
def run_two(n):
    for _ in range(n): ga.replay(); gb.replay()
res = {"one": [], "two": []}
for rnd in range(3):
    for name, fn in (("one", run_one), ("two", run_two)):
        fn(60)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(100); e.record(); torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 100)
for k, v in res.items():
    v = sorted(v); print(f"{k} graph(s): median {v[1]:.3f} ms  min {v[0]:.3f}")
