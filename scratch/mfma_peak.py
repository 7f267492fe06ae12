import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "mfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
blocks = 256 * 2   # 2 blocks of 4 waves per CU = 2 waves per SIMD
out = torch.empty(blocks * 256, device="cuda"); clk = torch.zeros(2, dtype=torch.int64, device="cuda")
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(iters):
    lib.mfma_peak(ctypes.c_void_p(out.data_ptr()), blocks, 10, ctypes.c_void_p(clk.data_ptr()), S); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.mfma_peak(ctypes.c_void_p(out.data_ptr()), blocks, iters, ctypes.c_void_p(clk.data_ptr()), S); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    flop = blocks * 4 * iters * 16 * 32768.0
    c = clk.tolist()
    print(f"iters {iters:8d}: {ms:9.3f} ms  {flop / ms / 1e9:8.1f} TFLOP/s   block 0: {c[0]} s_memtime ticks, {c[1]} wall ticks (100 MHz) -> s_memtime at {c[0] / max(c[1], 1) * 100:.0f} MHz; "
          f"MFMA issue clock if back to back: {blocks // 256 * 4 / 4 * iters * 16 * 32 / 2 / max(c[1], 1) * 100:.0f} MHz", flush=True)
for it in (200, 2000, 20000, 200000, 2000, 200):
    run(it)
