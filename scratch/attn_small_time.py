import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
B, H, S, hd = 16, 12, 77, 64
C = H * hd
qkv = torch.randn(B * S, 3 * C, device="cuda").half(); q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda")
ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd, causal=True)
do = torch.randn(B * S, C, device="cuda").half(); delta = torch.empty(B, H, S, device="cuda"); dqkv = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16)
def run(): ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, causal=True)
def timeit(reps=20):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): run()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
for bits in (1, 1 | 8192, 1):
    L.lib().tb_attention_set_variant(bits)
    print(f"variant {bits}: {timeit():.1f} us", flush=True)
L.lib().tb_attention_set_variant(1)
