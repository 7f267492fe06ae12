import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
step, _ = build_step()
step.capture(warmup=2)
g1 = step.graph[0]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    step.draw(); step.forward_backward(); step.optimizer_step()
torch.cuda.synchronize()
for i in range(6): g1.replay()
torch.cuda.synchronize()
time.sleep(0.2)
for i in range(6): (g1 if i % 2 == 0 else g2).replay()
torch.cuda.synchronize()
