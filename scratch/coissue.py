import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "coissue.so"))
out = torch.empty(256 * 512, device="cuda")
def run(mode, iters=2000, blocks=256):
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.coissue(mode, ctypes.c_void_p(out.data_ptr()), blocks, 10, S); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); lib.coissue(mode, ctypes.c_void_p(out.data_ptr()), blocks, iters, S); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us per iteration
names = ["0 homogeneous (MFMA then VALU in every wave)", "1 specialised (4 waves MFMA only, 4 waves VALU only)", "2 ping-pong (barrier-staggered groups)",
         "3 MFMA only", "4 VALU only", "5 specialised by wave parity", "6 specialised by wave bit 1", "7 specialised: MFMA | plain VALU", "8 specialised: MFMA | exp",
         "9 plain VALU only", "10 exp only"]
for m in range(11):
    t = run(m)
    print(f"mode {names[m]:60s}: {t*1e3:8.1f} ns / iteration / wave-pair-per-SIMD")
