"""Do the branches of ONE HIP graph run concurrently on this ROCm?  Two independent chains of small GEMMs (each far below one chip round)
captured (a) one after the other on one stream, (b) as a fork / join over two streams; also both chains eagerly on two streams.
And: how much of the ~0.3 ms between two replays of the step graph does a graph holding SEVERAL steps remove?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
from textboost_amd.workload import build_step

dev = "cuda"
M, N, K, L = 1024, 768, 768, 40      # 8 x 6 tiles of 128 x 128: 48 workgroups on 256 CUs
mk = lambda *s: torch.randn(*s, device=dev).half()
chains = []
for c in range(2):
    chains.append(dict(x=[mk(M, K), mk(M, N)], w=mk(N, K)))


def run_chain(c):
    a, b = c["x"]
    for i in range(L):
        ops.gemm(a, c["w"], b, alpha=1e-2)
        a, b = b, a


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run_chain(chains[0]); run_chain(chains[1]); torch.cuda.synchronize()
side = torch.cuda.Stream()
g_serial, g_fork = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g_serial):
    run_chain(chains[0]); run_chain(chains[1])
with torch.cuda.graph(g_fork):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side), ops.workspace_slot(1):
        run_chain(chains[1])
    run_chain(chains[0])
    main.wait_stream(side)
g_one = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_one):
    run_chain(chains[0])
print(f"one chain, graph:                {timed(g_one.replay):.3f} ms", flush=True)
print(f"two chains serial, one graph:    {timed(g_serial.replay):.3f} ms", flush=True)
print(f"two chains fork/join, one graph: {timed(g_fork.replay):.3f} ms", flush=True)
g_b = torch.cuda.CUDAGraph()
with torch.cuda.graph(g_b):
    run_chain(chains[1])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def two_graphs():
    with torch.cuda.stream(sa): g_one.replay()
    with torch.cuda.stream(sb): g_b.replay()


print(f"two chains, two graphs/streams:  {timed(two_graphs):.3f} ms", flush=True)

# ---- several steps per graph
step, _ = build_step()
step.capture(warmup=2)
g1 = step.graph[0]
ms1 = timed(g1.replay, 40)
print(f"step graph, 1 step per replay:   {ms1:.3f} ms/step", flush=True)
for n in (2, 4):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(n):
            step.draw(); step.forward_backward(); step.optimizer_step()
    ms = timed(g.replay, 40 // n) / n
    print(f"step graph, {n} steps per replay:  {ms:.3f} ms/step", flush=True)
print(step.scalars())
