"""graph replays of the step with tb_gemm8_set(bits) (argv[1]) for rocprofv3 --kernel-trace --stats: in-step durations of the convolution tile"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textboost_amd.workload import build_step
from textboost_amd import _lib as L
step, _ = build_step(batch=8, latent=64, data_seed=1000, world_size=1, device=torch.device("cuda", 0))
L.lib().tb_gemm8_set(int(sys.argv[1]))
for _ in range(2): step.step_eager()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step.draw(); step.forward_backward(); step.optimizer_step()
for _ in range(60): g.replay()
torch.cuda.synchronize()
