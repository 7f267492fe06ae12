import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "ldsmfma.so"))
out = torch.empty(256 * 512, device="cuda"); cyc = torch.zeros(4, dtype=torch.int64, device="cuda")
names = ["4 waves 128x80: 40 MFMAs only", "4 waves 128x80: 13 reads only", "4 waves 128x80: reads between MFMAs (pitch 144)",
         "8 waves 64x80: 20 MFMAs only", "8 waves 64x80: 9 reads only", "8 waves 64x80: reads between MFMAs (pitch 144)",
         "4 waves 128x80: MFMAs + reads + barrier per tap", "4 waves 128x80: MFMAs + reads + barrier + 6 DMA pieces per tap",
         "8 waves 64x80: MFMAs + reads + barrier per tap", "8 waves 64x80: MFMAs + reads + barrier + 3 DMA pieces per tap",
         "8 waves 64x80: MFMAs + barrier + 3 DMA pieces (no reads)", "4 waves 128x80: MFMAs + barrier + 6 DMA pieces (no reads)",
         "8 waves 64x80: MFMAs + reads + barrier + 3 global_load_dwordx4 -> ds_write_b128 per tap", "4 waves 128x80: the same with 6 per tap"]
gsrc = torch.randn(4 * 1024 * 1024, device="cuda").half()
S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for fill in (0, 4):
  for m, n in enumerate(names):
    iters = 10000
    rc = lib.ldsmfma(m, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), 256, 10, fill, ctypes.c_void_p(gsrc.data_ptr()), S); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); lib.ldsmfma(m, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(cyc.data_ptr()), 256, iters, fill, ctypes.c_void_p(gsrc.data_ptr()), S); e.record(); torch.cuda.synchronize()
    c = cyc[0].item() / (iters * 2)
    print(f"fill {fill} clock {cyc[0].item() / cyc[1].item() * 0.1:5.2f} GHz {n:52s}: {s.elapsed_time(e) * 1e6 / (iters * 2):7.1f} ns per k=32 step of the 256 x 160 tile ({c:7.1f} s_memtime ticks)")
