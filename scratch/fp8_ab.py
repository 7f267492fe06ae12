"""fp8 P.V forward vs the fp16 forward of the L0 self-attention shape (isolated, back to back), and the whole step with / without it"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
from textboost_amd.workload import build_step
def b2b(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B in (8, 16):
    H, S, hd = 8, 4096, 40; C = H * hd
    qkv = torch.randn(B * S, 3 * C, device="cuda").half()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    o = torch.empty(B * S, C, device="cuda", dtype=torch.float16); lse = torch.empty(B, H, S, device="cuda")
    ws = ops.attention_fp8_workspace(B, H, S, "cuda")
    t16 = b2b(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd))
    t8 = b2b(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd, fp8_ws=ws))
    print(f"B={B} S=4096 hd=40 forward: fp16 {t16:.1f} us, fp8 P.V incl. the two V-image passes {t8:.1f} us")
for B in (8, 16):
    res = {}
    for fp8 in (False, True):
        step, _ = build_step(batch=B, latent=64, attn_fp8=fp8)
        step.capture()
        for _ in range(3): step.replay()
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): step.replay()
        e.record(); torch.cuda.synchronize()
        res[fp8] = s.elapsed_time(e) / 20
        sc = step.scalars()
        print(f"B={B} fp8={fp8}: {res[fp8]:.3f} ms/step ({1000 / res[fp8]:.2f} steps/s), loss_mse {sc['loss_mse']:.4f}")
        del step
        torch.cuda.empty_cache()
