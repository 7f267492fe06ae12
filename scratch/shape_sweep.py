"""Every distinct tb_gemm launch of the step, timed COLD (a 640 MB fill between launches evicts L2 / MALL, as in the step) under the dispatcher's
default choice and under the tile / split / wide-tile overrides: which shapes does the default rule mis-dispatch?
usage: shape_sweep.py [min_count]"""
import sys, os, ctypes, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
from textboost_amd.workload import build_step
lib = L.lib()
orig = ops.gemm
REC, ARGS = [], {}
def gemm(A, W, out, **kw):
    if ops._REC is not None:
        M, N = out.shape[0], W.shape[0]
        conv = kw.get("conv")
        K = (d := conv) and {2: 4, 3: 16}.get(conv.get("upsample", 0), 9) * conv["Cin"] or W.shape[1] + (kw["W2"].shape[1] if kw.get("W2") is not None else 0)
        tag = ("conv%dx%d" % (conv["Hout"], conv["Wout"]) if conv else "lin") + ("/up%d" % conv["upsample"] if conv and conv.get("upsample") else "") + \
              ("/T" if conv and conv.get("transposed") else "") + ("/s2" if conv and conv.get("stride") == 2 else "") + \
              ("/act%d" % kw["act"] if kw.get("act") else "") + ("/ln" if kw.get("ln_fwd") or kw.get("ln_bwd") else "") + ("/R" if kw.get("R") is not None else "") + \
              ("/defer" if kw.get("defer") else "")
        key = (tag, M, N, K)
        REC.append(key); ARGS.setdefault(key, (A, W, out, dict(kw)))
    return orig(A, W, out, **kw)
ops.gemm = gemm
import textboost_amd.unet, textboost_amd.text_encoder
step, _ = build_step()
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
ops.start_recording(); step.step_eager(); torch.cuda.synchronize(); ops.stop_recording()
cnt = collections.Counter(REC)
flush = torch.empty(640 << 20, device="cuda", dtype=torch.uint8)
def time_cold(fn, reps=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            flush.fill_(1); fn()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0):
        for _ in range(reps): flush.fill_(1)
    ts = []
    for gg in (g, g0):
        gg.replay(); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); gg.replay(); gg.replay(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (2 * reps) * 1e3)
    return ts[0] - ts[1]
def cfg_name():
    cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
    c8 = (ctypes.c_int * 6)()
    if lib.tb_gemm8_last(c8): return "g8<%d,%d,%d,%d,%d,%d>" % tuple(c8)
    if cfg[2] == 2: return "halo<%d> S%d" % (cfg[1], cfg[4])
    return "%dx%d st%d S%d" % (cfg[0], cfg[1], cfg[3] % 10, cfg[4])
G8 = lib.tb_gemm8_set(39); lib.tb_gemm8_set(G8)
variants = [("default", [], [], None), ("64x64", [8001], [8000], None), ("128x64", [8002], [8000], None), ("128x128", [8003], [8000], None),
            ("nosplit", [1000], [1384], None), ("split512", [1512], [1384], None), ("no-g8", [], [], 0), ("g8 64x320", [], [], G8 | 8), ("g8 no128x160", [], [], G8 | 128)]
mincount = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rows = []
for key, n in cnt.most_common():
    if n < mincount: continue
    A, W, out, kw = ARGS[key]
    kw = {k: v for k, v in kw.items() if k != "defer"}     # (deferred split-K slices need their consumer: time the reducer form)
    res = []
    for name, codes, reset, g8 in variants:
        for c in codes: lib.tb_gemm_set_variant(c)
        if g8 is not None: lib.tb_gemm8_set(g8)
        try:
            t = time_cold(lambda: orig(A, W, out, **kw)); nm = cfg_name()
        except Exception as ex:
            t, nm = float("nan"), "err"
        for c in reset: lib.tb_gemm_set_variant(c)
        if g8 is not None: lib.tb_gemm8_set(G8)
        res.append((name, t, nm))
    d = res[0][1]
    best = min((r for r in res if r[1] == r[1]), key=lambda r: r[1])
    rows.append(((d - best[1]) * n, key, n, res, best))
rows.sort(key=lambda r: -r[0])
print("total potential: %.3f ms per step" % (sum(r[0] for r in rows) / 1e3))
for gain, key, n, res, best in rows[:45]:
    print(f"{key[0]:28s} M={key[1]:6d} N={key[2]:6d} K={key[3]:6d} x{n:3d}  default {res[0][1]:6.1f} us [{res[0][2]}]  best {best[0]} {best[1]:6.1f} [{best[2]}]  -> {gain / 1e3:6.3f} ms")
    print("      " + "  ".join(f"{nm}:{t:.1f}" for nm, t, _ in res))
