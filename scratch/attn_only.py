import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
S, Skv, hd = (4096, 4096, 40) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1].split(","))
B, H = 8, 8; C = H * hd
q = torch.randn(B * S, C, device=dev).half(); k = torch.randn(B * Skv, C, device=dev).half(); v = torch.randn(B * Skv, C, device=dev).half()
o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev); do = torch.randn_like(q); delta = torch.empty_like(lse)
dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
for _ in range(3):
    ops.attention_fwd(q, k, v, o, lse, B, H, S, Skv, hd)
    ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, S, Skv, hd)
torch.cuda.synchronize()
