"""cross-attention backward: one-launch kernel (round 5) against the three launches of round 4 (variant bit 65536), graph-timed"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
for B, H, Sq, hd in ((8, 8, 4096, 40), (8, 8, 1024, 80), (8, 5, 9216, 64), (8, 10, 2304, 64), (8, 20, 576, 64)):
    Skv, C = 77, H * hd
    q = torch.randn(B * Sq, C, device="cuda").half(); kv = torch.randn(B * Skv, 2 * C, device="cuda").half(); do = torch.randn(B * Sq, C, device="cuda").half()
    o = torch.empty_like(q); lse = torch.empty(B, H, Sq, device="cuda"); delta = torch.empty(B, H, Sq, device="cuda")
    dq = torch.empty_like(q); dkv = torch.empty_like(kv); ws = torch.empty(16 * 2 * B * Skv * C, device="cuda")
    ops.attention_fwd(q, kv[:, :C], kv[:, C:], o, lse, B, H, Sq, Skv, hd)
    out = []
    for var in (1, 1 | 65536, 1, 1 | 65536):
        L.lib().tb_attention_set_variant(var)
        out.append(timeit(lambda: ops.attention_bwd(q, kv[:, :C], kv[:, C:], o, lse, do, delta, dq, dkv[:, :C], dkv[:, C:], B, H, Sq, Skv, hd, ws=ws)))
    L.lib().tb_attention_set_variant(1)
    print(f"B={B} H={H} Sq={Sq} hd={hd}: one launch {out[0]:.1f} / {out[2]:.1f} us, three launches {out[1]:.1f} / {out[3]:.1f} us")
