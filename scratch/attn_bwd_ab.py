"""L0 self-attention backward (B=8, H=8, S=4096, hd=40): register-staged vs LDS-DMA staged dK/dV kernel, back to back"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
B, H, S, hd = 8, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 40; C = H * hd
qkv = torch.randn(B * S, 3 * C, device="cuda").half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda")
ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
do = torch.randn(B * S, C, device="cuda").half(); delta = torch.empty(B, H, S, device="cuda")
dqkv = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16); ws = torch.empty(2 * B * H * S, device="cuda")
def run(): ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
for name, bits in [("warm-up", 1), ("dma dq + dkv", 1), ("register-staged", 1 | 128 | 256), ("dma dkv only", 1 | 256), ("dma dq only", 1 | 128), ("dma + remap", 1 | 64), ("dma dq + dkv", 1)]:
    L.lib().tb_attention_set_variant(bits)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    print(f"{name:20s} bwd (dq + dkv) {s.elapsed_time(e) / 10 * 1e3:8.1f} us")
L.lib().tb_attention_set_variant(1)
