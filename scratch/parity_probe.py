"""actual errors of the HIP UNet vs the fp32 oracle and vs the fp16-faithful oracle (tiny config): rel-L2, max-abs, per-channel"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_model as T
from oracle.fp16_mode import fp16_rounding
dev = "cuda"
def stats(name, a, b, ch_dim=None):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    rel = ((a - b).norm() / b.norm()).item(); mx = (a - b).abs().max().item(); sc = b.abs().max().item()
    s = f"{name}: rel-L2 {rel:.3e} max-abs {mx:.3e} (ref max {sc:.3e}, ratio {mx/sc:.3e})"
    if ch_dim is not None:
        dims = [d for d in range(a.dim()) if d != ch_dim]
        pc = ((a - b).pow(2).sum(dims).sqrt() / b.pow(2).sum(dims).sqrt().clamp_min(1e-12))
        s += f" worst-channel rel {pc.max().item():.3e}"
    print(s)
for sd2 in (False, True):
    B, hw, D = 2, 16, 64
    ref, hip, cfg = T.make_unet(B, hw, D, sd2=sd2)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 4, hw, hw, generator=g).half().float(); t = torch.tensor([17, 801])
    ehs0 = torch.randn(B, 77, D, generator=g).half().float()
    dpred = torch.randn(B, 4, hw, hw, generator=g)
    res = {}
    for mode in ("fp32", "fp16"):
        ehs = ehs0.clone().requires_grad_(True)
        if mode == "fp16":
            with fp16_rounding():
                pr = ref(x, t, ehs); pr.backward(dpred.half().float())
        else:
            pr = ref(x, t, ehs); pr.backward(dpred)
        res[mode] = (pr.detach(), ehs.grad.clone())
    pred = hip.forward(x.half().to(dev), t.to(dev), ehs0.half().view(B * 77, D).to(dev).contiguous())
    d_ehs = hip.backward(dpred.to(dev)).view(B, 77, D)
    print("sd2" if sd2 else "sd1", "tiny")
    for mode in ("fp32", "fp16"):
        stats(f"  pred  vs {mode} oracle", pred, res[mode][0], 1)
        stats(f"  d_ehs vs {mode} oracle", d_ehs, res[mode][1], 2)
    stats("  fp16 oracle vs fp32 oracle pred", res["fp16"][0], res["fp32"][0], 1)
    stats("  fp16 oracle vs fp32 oracle d_ehs", res["fp16"][1], res["fp32"][1], 2)
