import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops
dev = "cuda"
M, D, r, P = 1232, 768, 4, 3
x = torch.randn(M, D, device=dev).half(); A = torch.randn(P * r, D, device=dev) / r; Bc = torch.randn(P * D, r, device=dev) * 0.1
t = torch.zeros(M, 64, device=dev, dtype=torch.float16); ops.lora_down(x, A, t)
dY = torch.randn(M, P * D, device=dev).half(); dt = torch.zeros(M, 64, device=dev, dtype=torch.float16)
dA = torch.zeros_like(A); dB = torch.zeros_like(Bc)
for _ in range(20): ops.lora_bwd(dY, x, t, Bc, dt, dA, dB, D, D, r, P)
torch.cuda.synchronize()
