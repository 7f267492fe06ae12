"""Do two step graphs replayed CONCURRENTLY on two streams fill each other's kernel-boundary bubbles?
Aggregate images/s of: one B=8 graph; two B=4 graphs side by side; two B=8 graphs side by side."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step


def make(batch, seed):
    step, _ = build_step(batch=batch, data_seed=seed)
    step.capture(warmup=2)
    return step


def run_one(step, n=40):
    g = step.graph[0]
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def run_two(a, b, n=40):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ga, gb = a.graph[0], b.graph[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(sa): ga.replay()
        with torch.cuda.stream(sb): gb.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s8 = make(8, 1000)
ms = run_one(s8)
print(f"one B=8 graph:            {ms:.3f} ms -> {8e3 / ms:.1f} images/s", flush=True)
s4a, s4b = make(4, 1000), make(4, 2000)
ms4 = run_one(s4a)
print(f"one B=4 graph:            {ms4:.3f} ms -> {4e3 / ms4:.1f} images/s", flush=True)
ms = run_two(s4a, s4b)
print(f"two B=4 graphs, 2 streams: {ms:.3f} ms -> {8e3 / ms:.1f} images/s", flush=True)
s8b = make(8, 2000)
ms = run_two(s8, s8b)
print(f"two B=8 graphs, 2 streams: {ms:.3f} ms -> {16e3 / ms:.1f} images/s", flush=True)
ms = run_one(s8)
print(f"one B=8 graph (again):    {ms:.3f} ms -> {8e3 / ms:.1f} images/s", flush=True)
