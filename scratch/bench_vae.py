import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import models, ops
from textboost_amd.vae import HipVAEEncoder, VAEGeometry, vae_encoder_shapes
dev = "cuda"
geo = VAEGeometry(); B = 8
sd = models.random_state_dict(vae_encoder_shapes(geo), 21, device=dev)
enc = HipVAEEncoder(geo, sd, B, 512, 512, device=dev)
x = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
eps = torch.randn(B, 4, 64, 64, device=dev)
for _ in range(2): enc.encode(x, noise=eps)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): enc.encode(x, noise=eps)
g.replay(); torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): g.replay()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"VAE encode B=8 512^2: {ms:.3f} ms  ({8 * 1.1e12 / ms / 1e9:.0f} TFLOP/s at ~1.1 TFLOP/img)")
ops.start_recording(); enc.encode(x, noise=eps); torch.cuda.synchronize(); rec = ops.stop_recording()
agg = collections.OrderedDict()
for name, fl, by, e0, e1 in rec:
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(v[1] for v in agg.values()); print("recorded ms", tot, "TFLOP", sum(v[2] for v in agg.values()) / 1e12)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:44s} x{v[0]:3d} {v[1]:8.3f} ms  {v[2]/v[1]/1e9 if v[1] else 0:8.1f} TF/s")
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
