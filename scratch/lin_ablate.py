"""Where the 20-24 us of the 6.7-GFLOP Linear shapes of levels 1 / 2 go: cold launches (rotating operands in a graph) of the 4-wave kernel
under its ablation knobs (tb_gemm_set_variant(2000 + bits): 1 = no k-loop loads, 2 = no MFMAs, 4 = no epilogue).  usage: lin_ablate.py"""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
lib = L.lib()
dev = "cuda"
WARM = os.environ.get('TB_WARM', '0')   # 1: four weight matrices, no flush (weights L2 / Infinity-Cache warm); 2: cold weights, ONE warm activation buffer
def bench(M, N, K, codes, reset, reps=5):
    nW = 4 if WARM == '1' else 64
    As = [torch.randn(M, K, device=dev).half() for _ in range(8)]
    Ws = [torch.randn(N, K, device=dev).half() * 0.05 for _ in range(nW)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(8)]
    big = torch.empty(600 << 20, device=dev, dtype=torch.uint8)   # flushed between replays: weights and activations come from HBM
    for c in codes: lib.tb_gemm_set_variant(c)
    if codes: old8 = lib.tb_gemm8_set(0); lib.tb_gemm_set_variant(9400)   # the 4-wave kernel (it has the ablation knobs)
    def run():
        for i in range(nW): ops.gemm(As[0 if WARM == '2' else i % 8], Ws[i], Os[i % 8])
    run(); torch.cuda.synchronize()
    cfg = (ctypes.c_int * 5)(); lib.tb_gemm_last_config(cfg)
    g8 = bool(lib.tb_gemm8_last(None))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): run()
    g.replay(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if WARM == '0': big.fill_(1)
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    for c in reset: lib.tb_gemm_set_variant(c)
    if codes: lib.tb_gemm8_set(old8); lib.tb_gemm_set_variant(9401)
    return tot / (reps * nW) * 1e3, ("g8 " if g8 else "") + f"{cfg[0]}x{cfg[1]} st{cfg[3] % 10} S{cfg[4]}"
import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == "stages":
    variants = [("default", [], []),
                ("64x64 st2", [8001, 10], [8000, 9]), ("64x64 st3", [8001, 14], [8000, 9]), ("64x64 st4", [8001, 15], [8000, 9]),
                ("128x64 st2", [8002, 0], [8000, 0]), ("128x64 st3", [8002, 4], [8000, 0]), ("128x64 st4", [8002, 5], [8000, 0]), ("128x64 st6", [8002, 6], [8000, 0]),
                ("128x128 st2", [8003, 0], [8000, 0]), ("128x128 st3", [8003, 4], [8000, 0]), ("128x128 st4", [8003, 5], [8000, 0])]
    shapes = [(2048, 1280, 1280), (8192, 640, 640), (2048, 1280, 2560), (8192, 640, 1920), (512, 1280, 1280)]
else:
    variants = None
no8 = [8001]   # force the 4-wave 64x64 tile (its ablation knobs)
_v0 = [("default", [], []),
            ("4-wave 64x64", [8001], [8000]), ("  no loads", [8001, 2001], [8000, 2000]), ("  no MFMA", [8001, 2002], [8000, 2000]),
            ("  no epilogue", [8001, 2004], [8000, 2000]), ("  no loads, no MFMA", [8001, 2003], [8000, 2000]), ("  nothing", [8001, 2007], [8000, 2000]),
            ("4-wave 128x128", [8003], [8000]), ("  no loads", [8003, 2001], [8000, 2000]), ("  no MFMA", [8003, 2002], [8000, 2000]), ("  nothing", [8003, 2007], [8000, 2000])]
if variants is None: variants, shapes = _v0, [(2048, 1280, 1280), (8192, 640, 640), (32768, 320, 320)]
for M, N, K in shapes:
    print(f"{M}x{N}x{K}", flush=True)
    for name, codes, reset in variants:
        t, cfg = bench(M, N, K, codes, reset)
        print(f"    {name:22s} {t:7.1f} us   {cfg}", flush=True)
