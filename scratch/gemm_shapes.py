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
        import ctypes
        cfg = (ctypes.c_int * 5)(); L.lib().tb_gemm_last_config(cfg)
        c8 = (ctypes.c_int * 6)()
        if L.lib().tb_gemm8_last(c8): kn = "g8<%d,%d,%d,%d,%d,%d>" % tuple(c8)
        elif cfg[2] == 2: kn = "halo<%d>" % cfg[1]
        else: kn = "gemm<%d,%d,m%d,ns%d,S%d>" % (cfg[0], cfg[1], cfg[2], cfg[3] % 10, cfg[4])
        REC.append((tag + " " + kn, M, N, K, e0, e1)); ARGS.setdefault((tag + " " + kn, M, N, K), (A, W, out, kw)); return r
    return orig(A, W, out, **kw)
ops.gemm = gemm
import textboost_amd.unet, textboost_amd.text_encoder
REC = []
ARGS = {}
step, _ = build_step()
for _ in range(3): step.step_eager()
torch.cuda.synchronize()
ops.start_recording(); step.step_eager(); torch.cuda.synchronize(); ops.stop_recording()
agg = collections.OrderedDict()
for tag, M, N, K, e0, e1 in REC:
    a = agg.setdefault((tag, M, N, K), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f}")
rep = {}
for key, (A, W, out, kw) in ARGS.items():
    for _ in range(3): orig(A, W, out, **kw)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): orig(A, W, out, **kw)
    e1.record(); torch.cuda.synchronize(); rep[key] = e0.elapsed_time(e1) / 20 * 1e3
print(f"sum over launches of back-to-back time: {sum(rep[k] * v[0] for k, v in agg.items()) / 1e3:.2f} ms")
for (tag, M, N, K), (n, t) in sorted(agg.items(), key=lambda kv: -rep[kv[0]] * kv[1][0])[:70]:
    print(f"b2b {rep[(tag, M, N, K)]:6.1f} us x{n:3d} = {rep[(tag, M, N, K)] * n / 1e3:6.3f} ms {2*M*N*K/rep[(tag, M, N, K)]/1e6:7.1f} TF | " + f"{tag:34s} M={M:6d} N={N:6d} K={K:6d}  x{n:3d}  {t:7.3f} ms  avg {t/n*1e3:7.1f} us  {2*M*N*K*n/t/1e9:7.1f} TF/s")
