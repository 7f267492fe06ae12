import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
from attn_il_test import ref_attn
dev = "cuda"
B, H, S, hd = 1, 8, 1024, 40; C = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * C, device=dev).half()
h = 0
kk = 519
qkv[kk, C + h * hd:C + (h + 1) * hd] = (qkv[5, h * hd:(h + 1) * hd].float() * 6).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
oref, lref = ref_attn(q, k, v, B, H, S, hd)
L.lib().tb_attention_set_variant(1)
ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd); torch.cuda.synchronize()
bad = ~torch.isfinite(o.float()).all(dim=1)
print("rows with non-finite O:", bad.nonzero().flatten().tolist()[:40], "count", int(bad.sum()))
ob = o.float().view(S, H, hd)
badh = ~torch.isfinite(ob).all(dim=2)
print("per head bad counts", badh.sum(0).tolist())
err = (ob - oref.view(S, H, hd)).abs().amax(dim=2)
print("rows (head 0) with err > 1e-2:", (err[:, 0] > 1e-2).nonzero().flatten().tolist()[:64])
print("lse err rows head0:", ((lse[0, 0] - lref[0, 0]).abs() > 1e-2).nonzero().flatten().tolist()[:64])
print("lse[0,0,0:8]", lse[0, 0, :8].tolist(), "ref", lref[0, 0, :8].tolist())
