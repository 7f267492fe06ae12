"""hd = 40 self-attention backward at the metric shape: 80-byte (no pad chunk) vs 96-byte LDS rows.  The dQ kernel switches at run time
(tb_attention_set_variant bit 8192 = padded rows); the dK/dV kernel is a build option (TB_LIB_SUFFIX=_alt TB_CFLAGS=-DTB_IL_DKV_PAD=1)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
B, H, S, hd = 8, 8, 4096, 40; C = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * C, device="cuda").half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda")
ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
do = torch.randn(B * S, C, device="cuda").half(); delta = torch.empty(B, H, S, device="cuda")
ws = torch.empty(2 * B * H * S, device="cuda")
def run(out): ops.attention_bwd(q, k, v, o, lse, do, delta, out[:, :C], out[:, C:2 * C], out[:, 2 * C:], B, H, S, S, hd, ws=ws)
base = L.lib().tb_attention_set_variant(0); L.lib().tb_attention_set_variant(base)
res = {}
outs = {}
for rnd in range(3):
    for name, bits in [("dq 80-byte rows", base), ("dq 96-byte rows", base | 8192)]:
        L.lib().tb_attention_set_variant(bits)
        out = torch.zeros(B * S, 3 * C, device="cuda", dtype=torch.float16)
        for _ in range(3): run(out)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run(out)
        e.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(s.elapsed_time(e) / 20 * 1e3)
        outs[name] = out
L.lib().tb_attention_set_variant(base)
for n, v_ in res.items(): print(f"[{os.environ.get('TB_LIB_SUFFIX', 'default lib')}] {n:18s} bwd (dq + dkv) median {sorted(v_)[1]:8.1f} us  min {min(v_):8.1f}")
a, b_ = outs["dq 80-byte rows"], outs["dq 96-byte rows"]
print("dq bit-equal:", torch.equal(a[:, :C], b_[:, :C]), " finite:", bool(torch.isfinite(a.float()).all()))
torch.save(a.cpu(), f"/tmp/attn_bwd_out{os.environ.get('TB_LIB_SUFFIX', '')}.pt")
if os.path.exists("/tmp/attn_bwd_out.pt") and os.path.exists("/tmp/attn_bwd_out_alt.pt"):
    x, y = torch.load("/tmp/attn_bwd_out.pt"), torch.load("/tmp/attn_bwd_out_alt.pt")
    print("dkv 80-byte vs 96-byte build bit-equal:", torch.equal(x, y))
