"""Is HIP graph launch host-bound at the start of each step?  host time per replay(); one exec vs two alternating execs."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
step, _ = build_step()
step.capture(warmup=2)
g1 = step.graph[0]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    step.draw(); step.forward_backward(); step.optimizer_step()
def run(graphs, n=20):
    for g in graphs: g.replay()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter(); graphs[i % len(graphs)].replay(); host.append(time.perf_counter() - h0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3, min(host) * 1e3, max(host) * 1e3
for name, gs in (("one exec", [g1]), ("two execs", [g1, g2]), ("one exec", [g1]), ("two execs", [g1, g2])):
    ms, hostms, hmin, hmax = run(gs)
    print(f"{name}: {ms:.3f} ms/step ({1e3/ms:.2f} steps/s); host loop {hostms:.3f} ms/step, replay() call min {hmin:.3f} max {hmax:.3f} ms", flush=True)
# replay + sync each step (no run-ahead)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): g1.replay(); torch.cuda.synchronize()
print(f"sync every step: {(time.perf_counter()-t0)/10*1e3:.3f} ms/step")
print(step.scalars())
