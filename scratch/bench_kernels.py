import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
print("== linear GEMMs (M,N,K)")
for M, N, K in [(32768, 320, 320), (32768, 2560, 320), (32768, 320, 1280), (8192, 640, 640), (8192, 5120, 640), (2048, 1280, 1280), (2048, 10240, 1280), (512, 1280, 1280), (616, 2304, 832), (616, 3072, 768), (616, 768, 3072), (616, 24960, 768), (32768, 960, 320)]:
    A = torch.randn(M, K, device=dev).half(); W = torch.randn(N, K, device=dev).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    t = timeit(lambda: ops.gemm(A, W, out))
    print(f"  {M:6d} {N:6d} {K:6d}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s")
print("== conv3x3 (B=8) Cin->Cout @ HxW")
for Ci, Co, H in [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (1280, 1280, 8), (2560, 1280, 8), (2560, 1280, 16), (1920, 640, 32), (960, 320, 64), (640, 320, 64), (320, 640, 32)]:
    B = 8
    x = torch.randn(B * H * H, Ci, device=dev).half(); w = torch.randn(Co, 9 * Ci, device=dev).half(); out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
    t = timeit(lambda: ops.gemm(x, w, out, conv=geo))
    print(f"  {Ci:5d}->{Co:5d} @{H:3d}: {t*1e6:8.1f} us  {2*B*H*H*Co*9*Ci/t/1e12:7.1f} TF/s")
print("== attention fwd / bwd (B=8,H=8)")
for S, Skv, hd in [(4096, 4096, 40), (1024, 1024, 80), (256, 256, 160), (64, 64, 160), (4096, 77, 40), (1024, 77, 80)]:
    B, H = 8, 8; C = H * hd
    q = torch.randn(B * S, C, device=dev).half(); k = torch.randn(B * Skv, C, device=dev).half(); v = torch.randn(B * Skv, C, device=dev).half()
    o = torch.empty_like(q); lse = torch.empty(B, H, S, device=dev); do = torch.randn_like(q); delta = torch.empty_like(lse)
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    t = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, Skv, hd))
    fl = 4 * B * H * S * Skv * hd
    tb = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, S, Skv, hd))
    print(f"  S={S:5d} Skv={Skv:5d} hd={hd:3d}: fwd {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s | bwd {tb*1e6:8.1f} us {2.5*fl/tb/1e12:6.1f} TF/s(alg 2.5x)")
print("== groupnorm fwd+bwd (B=8)")
for HW, C in [(4096, 320), (4096, 960), (1024, 640), (256, 1280), (64, 2560)]:
    B = 8; M = B * HW
    x = torch.randn(M, C, device=dev).half(); y = torch.empty_like(x); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    st = torch.empty(B, 32, 2, device=dev); ws = torch.empty(ops.groupnorm_ws(B, HW, C), device=dev); dy = torch.randn_like(x); dx = torch.empty_like(x)
    t = timeit(lambda: ops.groupnorm_fwd(x, y, g, b, st, ws, B, HW, C, silu=True))
    tb = timeit(lambda: ops.groupnorm_bwd(dy, x, g, b, st, dx, ws, B, HW, C, silu=True))
    print(f"  HW={HW:5d} C={C:5d}: fwd {t*1e6:7.1f} us ({3*M*C*2/t/1e9:6.0f} GB/s alg)  bwd {tb*1e6:7.1f} us ({5*M*C*2/tb/1e9:6.0f} GB/s alg)")
print("== layernorm fwd (M,C)")
for M, C in [(32768, 320), (8192, 640), (2048, 1280), (616, 768)]:
    x = torch.randn(M, C, device=dev).half(); y = torch.empty_like(x); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); st = torch.empty(M, 2, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, y, g, b, st))
    print(f"  {M:6d} {C:5d}: {t*1e6:7.1f} us ({2*M*C*2/t/1e9:6.0f} GB/s)")
