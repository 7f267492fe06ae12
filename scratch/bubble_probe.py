"""Is the ~0.1-0.2 ms submission bubble near node 16 of every replay real without the tracer?  Stamp kernels (wall_clock64, 100 MHz) in front of the
first N gemm / groupnorm / attention calls of the captured step; per-interval medians over replays."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd.workload import build_step
from textboost_amd import ops
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "stamp.so"))
NS = int(os.environ.get("NSTAMP", "40"))
buf = torch.zeros(NS + 2, dtype=torch.int64, device="cuda")
state = {"n": 0, "on": False, "names": []}
def stamp(name):
    if state["on"] and state["n"] < NS:
        lib.run_stamp(ctypes.c_void_p(buf.data_ptr() + 8 * state["n"]), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        state["names"].append(name); state["n"] += 1
for fn in ("gemm", "groupnorm_fwd", "attention_fwd", "conv4_to_nhwc", "layernorm_fwd", "add_noise", "timestep_embed", "ff_fwd"):
    if hasattr(ops, fn):
        orig = getattr(ops, fn)
        setattr(ops, fn, (lambda o, n: (lambda *a, **k: (stamp(n), o(*a, **k))[1]))(orig, fn))
step, _ = build_step()
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
state["on"] = True; state["n"] = 0; state["names"] = []
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step.draw(); step.forward_backward(); step.optimizer_step()
    lib.run_stamp(ctypes.c_void_p(buf.data_ptr() + 8 * (NS + 1)), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
state["on"] = False
rows = []
for _ in range(60): g.replay()
for _ in range(30):
    g.replay(); torch.cuda.synchronize()
    rows.append(buf.tolist())
import statistics
n = state["n"]
print("stamped", n, "calls; step (first stamp -> end)", statistics.median((r[NS + 1] - r[0]) / 100.0 for r in rows), "us")
for i in range(n - 1):
    d = [(r[i + 1] - r[i]) / 100.0 for r in rows]
    print(f"{i:3d} {state['names'][i]:16s} -> next: median {statistics.median(d):8.1f} us  max {max(d):8.1f}")
