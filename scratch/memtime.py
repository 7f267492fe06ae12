import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "memtime.so"))
out = torch.zeros(4, dtype=torch.int64, device="cuda")
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ticks in (10_000_000, 200_000_000):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); lib.run_spin(ctypes.c_ulonglong(ticks), ctypes.c_void_p(out.data_ptr()), S); e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    print(f"ticks {out[0].item()} clock64 {out[1].item()} in {ms:.3f} ms -> s_memtime {out[0].item()/ms/1e3:.1f} MHz, clock64 {out[1].item()/ms/1e3:.1f} MHz")
