"""power / clock of the board while the step graph replays back to back (is the sustained step at the power cap?)"""
import os, sys, subprocess, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step.draw(); step.forward_backward(); step.optimizer_step()
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.split(":", 1)[1].strip() if False else l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "Power (W)", "junction", "fclk"))]
    return " | ".join(k.replace("GPU[0]", "").replace("\t", "").strip(": ") for k in keep)
print("idle:", smi(), flush=True)
t0 = time.time()
while time.time() - t0 < 25:
    for _ in range(40): g.replay()      # ~1.1 s of queued work
    print(f"t={time.time() - t0:5.1f}s", smi(), flush=True)
    torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): g.replay()
e.record(); torch.cuda.synchronize()
print("ms per step", s.elapsed_time(e) / 100)
