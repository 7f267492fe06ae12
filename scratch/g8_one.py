"""one conv shape through gemm8 a few times (for rocprofv3 --pmc runs)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; B = 8
Ci, Co, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(B * H * H, Ci, device=dev).half(); w = (torch.randn(Co, 9 * Ci, device=dev) / (9 * Ci) ** 0.5).half()
out = torch.empty(B * H * H, Co, device=dev, dtype=torch.float16)
geo = dict(B=B, Hin=H, Win=H, Cin=Ci, Hout=H, Wout=H, stride=1, sign=1, upsample=0, transposed=0)
for _ in range(6): ops.gemm(x, w, out, conv=geo)
torch.cuda.synchronize()
