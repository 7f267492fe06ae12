import sys, os, json, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
from textboost_amd.workload import build_step
# monkeypatch gemm to record shapes
orig = ops.gemm
def gemm(A, W, out, **kw):
    if ops._REC is not None:
        M, N = out.shape[0], W.shape[0]
        conv = kw.get("conv")
        K = 9 * conv["Cin"] if conv else W.shape[1] + (kw["W2"].shape[1] if kw.get("W2") is not None else 0)
        tag = ("conv" if conv else "lin") + ("/up" if conv and conv["upsample"] else "") + ("/T" if conv and conv["transposed"] else "") + ("/s2" if conv and conv["stride"] == 2 else "") + ("/geglu" if kw.get("act") == L.ACT_GEGLU else "")
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(A, W, out, **kw); e1.record()
        REC.append((tag, M, N, K, e0, e1)); return r
    return orig(A, W, out, **kw)
ops.gemm = gemm
import textboost_amd.unet, textboost_amd.text_encoder
REC = []
step, _ = build_step()
for _ in range(3): step.step_eager()
torch.cuda.synchronize()
ops.start_recording(); step.step_eager(); torch.cuda.synchronize(); ops.stop_recording()
agg = collections.OrderedDict()
for tag, M, N, K, e0, e1 in REC:
    a = agg.setdefault((tag, M, N, K), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f}")
for (tag, M, N, K), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{tag:12s} M={M:6d} N={N:6d} K={K:6d}  x{n:3d}  {t:7.3f} ms  avg {t/n*1e3:7.1f} us  {2*M*N*K*n/t/1e9:7.1f} TF/s")
