import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
from textboost_amd.sampler import DPMSolverPP2M
sch = DPMSolverPP2M(); sch.set_timesteps(25)
dev = "cuda"; torch.manual_seed(0)
B, n, g = 2, 4 * 16 * 16, 7.5
x = torch.randn(B, n, device=dev); m_prev = torch.zeros(B, n, device=dev); x2 = torch.zeros(2 * B, n, device=dev, dtype=torch.float16)
for i in range(25):
    e = (torch.randn(2 * B, n, device=dev) * 0.8).half()
    a_t, s_t = sch.alpha_sigma(sch.sigmas[i])
    ops.dpm_step(x, e, m_prev, x2, n, B, g, a_t, s_t, *sch.coefficients(i))
    d = (x2[:B].float() - x.half().float()).abs()
    print(i, "x2 vs x.half: max", d.max().item(), "n diff", (d > 0).sum().item(), "x2[B:] equal x2[:B]", torch.equal(x2[:B], x2[B:]))
    if d.max() > 0: j = d.argmax(); print(x.view(-1)[j].item(), x2.view(-1)[j].item(), x.half().view(-1)[j].item(), torch.isnan(x).any().item())
