"""L0 / L1 self-attention backward (dq + dkv launches), graph-timed; library by TB_LIB_SUFFIX"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
out = []
for B, H, S, hd in ((8, 8, 4096, 40), (8, 8, 1024, 80)):
    C = H * hd
    qkv = torch.randn(B * S, 3 * C, device="cuda").half(); do = torch.randn(B * S, C, device="cuda").half()
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda"); delta = torch.empty(B, H, S, device="cuda")
    dqkv = torch.empty_like(qkv); ws = torch.empty(2 * B * H * S, device="cuda")
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
    f = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
    bw = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws))
    out.append(f"S={S} hd={hd}: fwd {f:.1f} us, bwd {bw:.1f} us")
print(os.environ.get("TB_LIB_SUFFIX", "base"), " | ".join(out))
