"""isolated timings at M = 32768 (one L0 transformer block at B = 8): the fused feed-forward alone, with its chained neighbours, and the launches they replace"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textboost_amd import ops, _lib as L
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_gemm import pack_geglu
def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * reps) * 1e3
M, C, INNER = 32768, 320, 1280
dev = "cuda"
torch.manual_seed(0)
w1 = (torch.randn(2 * INNER, C, device=dev) / C ** 0.5).half(); b1 = torch.randn(2 * INNER, device=dev) * 0.3
w2 = (torch.randn(C, INNER, device=dev) / INNER ** 0.5).half(); b2 = torch.randn(C, device=dev) * 0.3
wpre = (torch.randn(C, C, device=dev) / C ** 0.5).half(); bpre = torch.randn(C, device=dev)
wpost = (torch.randn(C, C, device=dev) / C ** 0.5).half(); bpost = torch.randn(C, device=dev)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
# rotate buffers so that activations come from HBM as in the step
NB = 6
bufs = [dict(o2=torch.randn(M, C, device=dev).half(), t1=torch.randn(M, C, device=dev).half(), xin=torch.randn(M, C, device=dev).half(),
             t2=torch.empty(M, C, device=dev, dtype=torch.float16), l3=torch.empty(M, C, device=dev, dtype=torch.float16), st=torch.empty(M, 2, device=dev),
             hg=torch.empty(M, 2 * INNER, device=dev, dtype=torch.float16), t3=torch.empty(M, C, device=dev, dtype=torch.float16),
             out=torch.empty(M, C, device=dev, dtype=torch.float16)) for _ in range(NB)]
w1p, b1p = pack_geglu(w1).contiguous(), pack_geglu(b1).contiguous()
k = [0]
def nxt():
    k[0] = (k[0] + 1) % NB
    return bufs[k[0]]
def pre_old():
    b = nxt(); ops.gemm(b["o2"], wpre, b["t2"], bias=bpre, R=b["t1"], ln_fwd=(gamma, beta, b["st"], b["l3"], 1e-5))
def ff_old():
    b = nxt(); ops.ff_fwd(b["l3"], w1p, b1p, w2, b2, b["hg"], b["t3"], R=b["t2"])
def post_old():
    b = nxt(); ops.gemm(b["t3"], wpost, b["out"], bias=bpost, R=b["xin"])
def chain(mode):
    def f():
        b = nxt()
        pre = (wpre, bpre, b["t1"], b["t2"], gamma, beta, b["st"], 1e-5) if mode & 1 else None
        post = (wpost, bpost, b["xin"], b["out"]) if mode & 2 else None
        ops.ff_fwd(b["o2"] if mode & 1 else b["l3"], w1p, b1p, w2, b2, b["hg"], None if mode & 2 else b["t3"], R=b["t2"], pre=pre, post=post)
    return f
for name, fn in (("attn2.to_out + LN3 (gemm8 LN epilogue)", pre_old), ("ff_fwd", ff_old), ("proj_out (lin320)", post_old),
                 ("ff_fwd + pre", chain(1)), ("ff_fwd + post", chain(2)), ("ff_fwd + pre + post", chain(3))):
    print(f"{name:44s} {timeit(fn):7.1f} us")
# ---- the Linear -> LayerNorm -> Linear pairs
for N2 in (960, 320):
    w2c = (torch.randn(N2, C, device=dev) / C ** 0.5).half()
    ys = [torch.empty(M, N2, device=dev, dtype=torch.float16) for _ in range(NB)]
    def old1():
        b = nxt(); ops.gemm(b["o2"], wpre, b["t2"], bias=bpre, R=b["t1"] if N2 == 320 else None, ln_fwd=(gamma, beta, b["st"], b["l3"], 1e-5))
    def old2():
        b = nxt(); ops.gemm(b["l3"], w2c, ys[k[0]])
    def new():
        b = nxt(); ops.chain320(b["o2"], wpre, bpre, b["t1"] if N2 == 320 else None, b["t2"], gamma, beta, b["st"], w2c, None, ys[k[0]])
    print(f"N2={N2}: Linear+LN epilogue {timeit(old1):6.1f} us + Linear {timeit(old2):6.1f} us   ->  chain320 {timeit(new):6.1f} us")
