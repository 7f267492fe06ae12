import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "coissue2.so"))
out = torch.empty(256 * 512, device="cuda")
def run(mode, iters=2000, blocks=256):
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.coissue2(mode, ctypes.c_void_p(out.data_ptr()), blocks, 10, S); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); lib.coissue2(mode, ctypes.c_void_p(out.data_ptr()), blocks, iters, S); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e6 / iters)
    return best
names = ["MFMA b2b (28) | idle", "idle | 168 fma", "MFMA b2b | 168 fma", "MFMA+nop8 | idle", "MFMA+nop8 | 168 fma", "MFMA+nop16 | idle", "MFMA+nop16 | 168 fma",
         "MFMA+4 fma fillers | idle", "MFMA+6 fillers | idle", "MFMA+8 fillers | idle", "both MFMA+6 fillers", "both MFMA+4 fillers", "both MFMA+3 fillers",
         "MFMA b2b | 168 fma prio3", "MFMA b2b prio3 | 168 fma", "MFMA b2b | 168 exp", "idle | 168 exp", "MFMA+4 exp fillers | idle", "MFMA+6 exp fillers | idle",
         "both MFMA b2b", "both MFMA+2 fillers", "MFMA b2b | 336 fma", "MFMA+nop16 | 336 fma", "MFMA+2 fillers | 336 fma"]
for m, n in enumerate(names):
    print(f"case {m:2d} {n:32s}: {run(m):8.1f} ns / iteration (28 MFMAs per MFMA wave)")
