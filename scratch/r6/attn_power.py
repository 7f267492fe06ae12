"""L0 / L1 self-attention forward and backward on N(0,1) operands and on ZERO operands: the instruction streams are data-independent, so a faster run
on zeros is clock (power), not cycles."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textboost_amd import ops, _lib as L
def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (2 * reps) * 1e3
for B, H, S, hd in ((8, 8, 4096, 40), (8, 8, 1024, 80)):
    C = H * hd
    for kind in ("randn", "zeros"):
        mk = (lambda *s: torch.randn(*s, device="cuda").half()) if kind == "randn" else (lambda *s: torch.zeros(*s, device="cuda").half())
        qkv, do = mk(B * S, 3 * C), mk(B * S, C)
        o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda"); delta = torch.empty(B, H, S, device="cuda")
        dqkv = torch.empty_like(qkv); ws = torch.empty(2 * B * H * S, device="cuda")
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
        f = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
        bw = timeit(lambda: ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws))
        print(f"S={S} hd={hd} {kind}: fwd {f:.1f} us, bwd {bw:.1f} us")
# the GEGLU projection and the plain Linear of the 64x64 maps
for M, N, K in ((32768, 2560, 320), (32768, 320, 320), (8192, 5120, 640)):
    for kind in ("randn", "zeros"):
        A = (torch.randn(M, K, device="cuda") if kind == "randn" else torch.zeros(M, K, device="cuda")).half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5 if kind == "randn" else torch.zeros(N, K, device="cuda")).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        print(f"Linear {M}x{N}x{K} {kind}: {timeit(lambda: ops.gemm(A, W, out)):.1f} us")
