"""A/B inside one process: the step graph with the text encoder split over graph branches vs all on the main stream."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None, "side prio", os.environ.get("TB_SIDE_PRIORITY"))
steps = {}
for name, split, conc in (("merged, fwd on main", False, -1), ("merged", False, 0), ("split, serial", True, 0), ("split, P side", True, 1), ("split, I side", True, 2), ("split, both", True, 3)):
    step, _ = build_step()
    step.split_te = split
    step.split_conc = max(conc, 0)
    if conc < 0: step.te_fwd_side = False
    step.capture(warmup=2)
    steps[name] = step
def run(step, n=40):
    for _ in range(3): step.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rnd in range(2):
    for name, st in steps.items():
        print(f"{name:22s}: {run(st):.3f} ms/step", flush=True)
