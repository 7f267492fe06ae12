import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma16.so"))
out = torch.empty(256 * 512, device="cuda"); cyc = torch.zeros(4, dtype=torch.int64, device="cuda")
names = ["32x32x16 b2b", "16x16x32 b2b", "16x16x16 b2b", "16x16x32/16x16x16 alternating", "32x32x8 b2b",
         "16x16x32 +1 fma", "16x16x32 +2 fma", "16x16x32 +3 fma", "16x16x32 +4 fma", "32x32x16 +4 fma", "32x32x16 +6 fma", "32x32x16 +8 fma",
         "16x16x32 +1 exp", "16x16x32 +2 exp", "32x32x16 +2 exp", "32x32x16 +4 exp", "16x16x16 +1 fma", "16x16x16 +2 fma"]
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for threads in (256, 512):
    for m, n in enumerate(names):
        iters = 2000
        lib.mfma16(m, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), 256, threads, 10, S); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); lib.mfma16(m, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), 256, threads, iters, S); e.record(); torch.cuda.synchronize()
        c = cyc[0].item() / (iters * 32)
        print(f"{threads // 64 // 4} wave/SIMD  {n:32s}: {c:6.1f} cycles per MFMA slot per wave   ({s.elapsed_time(e) * 1e6 / (iters * 32):6.2f} ns)")
