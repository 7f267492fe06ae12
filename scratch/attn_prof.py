import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; S, hd, B, H = 4096, 40, 8, 8; C = H * hd
qkv = torch.randn(B * S, 3 * C, device=dev).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
dbg = torch.zeros(64, device=dev)
for var in (1, 33):
    L.lib().tb_attention_set_variant(var)
    d = ops._attn_desc(q, k, v, o, lse, B, H, S, S, hd, hd ** -0.5, False)
    d.Delta = L.ptr(dbg)
    for _ in range(3): L.check(L.lib().tb_attention_fwd(d, L.stream()), "x")
    torch.cuda.synchronize()
    r = dbg[:20].view(4, 5).tolist()
    print("variant", var, "clocks per tile [QK+softmax+trwait, PV issue, tail->vmcnt wait, barrier] per wave:")
    for w in r: print("   ", [round(x) for x in (w[0], w[1], w[2], w[3], w[4])], "sum", round(sum(w)))
