"""Does alternating the replay stream hide the ~0.3 ms the runtime needs between two launches of the step graph?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
step, _ = build_step()
step.capture(warmup=2)
g1 = step.graph[0]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, capture_error_mode="thread_local"):
    step.draw(); step.forward_backward(); step.optimizer_step()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "one stream":
        for i in range(n): g1.replay()
    elif mode == "two streams, one exec":
        for i in range(n):
            s, prev = (sA, sB) if i % 2 == 0 else (sB, sA)
            s.wait_stream(prev)
            with torch.cuda.stream(s): g1.replay()
    else:
        for i in range(n):
            s, prev = (sA, sB) if i % 2 == 0 else (sB, sA)
            s.wait_stream(prev)
            with torch.cuda.stream(s): (g1 if i % 2 == 0 else g2).replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for mode in ("one stream", "two streams, one exec", "two streams, two execs", "one stream", "two streams, one exec", "two streams, two execs"):
    ms = run(mode)
    print(f"{mode:26s} {ms:.3f} ms/step ({1e3 / ms:.2f} steps/s)", flush=True)
print(step.scalars())
