import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; S, hd, B, H = 4096, 40, 8, 8; C = H * hd
var = int(sys.argv[1])
qkv = torch.randn(B * S, 3 * C, device=dev).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
L.lib().tb_attention_set_variant(var)
for _ in range(5): ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
torch.cuda.synchronize()
