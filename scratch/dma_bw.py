"""bytes/clock/CU of LDS-DMA vs register loads (scratch/dma_bw.hip)"""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dma_bw.so"))
lib.dma_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
src = torch.randn(256 << 20 >> 2, device="cuda")  # 256 MB
sink = torch.zeros(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
names = {0: "global_load_lds_dwordx4", 1: "global_load_dwordx4 -> VGPR", 2: "global_load_dwordx4 + ds_write_b128"}
for span_mb, label in [(2, "L2-resident (2 MB)"), (64, "MALL-resident (64 MB)"), (256, "HBM stream (256 MB)")]:
    for mode in (0, 1, 2):
        iters = 4096
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lib.dma_run(mode, src.data_ptr(), span_mb << 20, iters, 256, sink.data_ptr(), st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        byts = 256 * 8 * iters * 1024
        print(f"{label:24s} {names[mode]:38s} {ms*1e3:8.1f} us  {byts/ms/1e9:6.2f} TB/s  {byts/256/(ms*1e-3*2.1e9):5.1f} B/clk/CU (at 2.1 GHz)")
