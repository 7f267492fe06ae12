"""A/B of tb_gemm8_set bits inside one process: capture the step under each setting, replay alternately"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
from textboost_amd import _lib as L
variants = [int(v) for v in sys.argv[1:]] or [39, 39 + 512]
steps = {}
for v in variants:
    L.lib().tb_gemm8_set(v)
    step, _ = build_step()
    step.capture(warmup=2)
    steps[v] = step
L.lib().tb_gemm8_set(39)
def run(step, n=40):
    for _ in range(3): step.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    for v, st in steps.items():
        print(f"g8_enable={v}: {run(st):.3f} ms/step", flush=True)
