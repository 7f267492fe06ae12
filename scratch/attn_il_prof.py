import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; S, hd, B, H = 4096, 40, 8, 8; C = H * hd
qkv = torch.randn(B * S, 3 * C, device=dev).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
dbg = torch.zeros(64 + 1024 * 8, device=dev)
L.lib().tb_attention_set_variant(1)
d = ops._attn_desc(q, k, v, o, lse, B, H, S, S, hd, hd ** -0.5, False)
d.Delta = L.ptr(dbg)
for _ in range(3): L.check(L.lib().tb_attention_fwd(d, L.stream()), "x")
torch.cuda.synchronize()
r = dbg[:20].view(4, 5).tolist()
print("clocks per tile [vmcnt wait, barrier, DMA issue, phase X, phase Y] per wave (s_memtime ticks):")
for w in r: print("   ", [round(x) for x in w], "sum", round(sum(w)))
print("whole block: shader ticks", dbg[20].item(), "realtime ticks (100 MHz)", dbg[21].item(), "-> effective shader clock %.0f MHz, block time %.1f us" % (dbg[20].item() / dbg[21].item() * 100, dbg[21].item() / 100))
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): L.check(L.lib().tb_attention_fwd(d, L.stream()), "x")
e.record(); torch.cuda.synchronize()
print("kernel time (prof build) %.1f us" % (s.elapsed_time(e) * 100))

import numpy as np
raw = dbg[64:].cpu().numpy().view(np.float64).reshape(1024, 4)
t0 = raw[:, 0].min()
st, en = (raw[:, 0] - t0) / 100, (raw[:, 1] - t0) / 100
dur = en - st
print("blocks: start min/max %.1f/%.1f us, end max %.1f us, duration mean %.1f min %.1f max %.1f us" % (st.min(), st.max(), en.max(), dur.mean(), dur.min(), dur.max()))
order = np.argsort(st)
print("start time percentiles (us):", [round(float(np.percentile(st, q)), 1) for q in (0, 25, 50, 75, 100)])
print("duration by start quartile:", [round(float(dur[order[i*256:(i+1)*256]].mean()), 1) for i in range(4)])
clk = raw[:, 2] / (dur * 100) * 100
print("effective clock MHz by start quartile:", [round(float(clk[order[i*256:(i+1)*256]].mean())) for i in range(4)])
hw = raw[:, 3].astype(np.int64)
cu = (hw & 0xffffffff) >> 8 & 0xf; se = (hw & 0xffffffff) >> 13 & 0x7; xcc = hw >> 32
ids = xcc * 1000 + se * 16 + cu
import collections
c = collections.Counter(ids.tolist())
print("distinct (xcc,se,cu):", len(c), "blocks per cu min/max", min(c.values()), max(c.values()))
# concurrency at mid time
for tq in (20, 60, 100, 150, 200, 250):
    print("t=%d us: running blocks %d" % (tq, int(((st <= tq) & (en > tq)).sum())))
