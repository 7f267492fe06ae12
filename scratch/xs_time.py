"""Level-0 cross-attention forward (8 x 8 heads x 4096 queries x 77 keys, hd 40) cold in a graph: attn_xs_fwd_kernel against the general kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib()
B, H, Sq, Skv, hd = 8, 8, 4096, 77, 40
C = H * hd
n = 12
qs = [torch.randn(B * Sq, C, device="cuda").half() for _ in range(n)]
kv = torch.randn(B * Skv, 2 * C, device="cuda").half()
os_ = [torch.empty(B * Sq, C, device="cuda", dtype=torch.float16) for _ in range(n)]
lse = torch.empty(B, H, Sq, device="cuda")
big = torch.empty(600 << 20, device="cuda", dtype=torch.uint8)
for name, var in (("general", 1 | 16384), ("short-key", 1)):
    lib.tb_attention_set_variant(var)
    def run():
        for i in range(n): ops.attention_fwd(qs[i], kv[:, :C], kv[:, C:], os_[i], lse, B, H, Sq, Skv, hd)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    tot = 0
    for _ in range(5):
        big.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    print(f"{name:10s} {tot / 5 / n * 1e3:6.1f} us per launch (42 MB of Q + O: {42e6 / (tot / 5 / n * 1e-3) / 1e12:.2f} TB/s)")
lib.tb_attention_set_variant(1)
