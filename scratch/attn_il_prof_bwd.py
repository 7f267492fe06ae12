import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import ops, _lib as L
dev = "cuda"; S, hd, B, H = 4096, 40, 8, 8; C = H * hd
qkv = torch.randn(B * S, 3 * C, device=dev).half()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
o = torch.empty(B * S, C, device=dev, dtype=torch.float16); lse = torch.empty(B, H, S, device=dev)
L.lib().tb_attention_set_variant(1 | 1024)   # forward: the unprofiled DMA kernel
ops.attention_fwd(q, k, v, o, lse, B, H, S, S, hd)
L.lib().tb_attention_set_variant(1)
do = torch.randn(B * S, C, device=dev).half(); delta = torch.zeros(B, H, S, device=dev)
dqkv = torch.zeros(B * S, 3 * C, device=dev, dtype=torch.float16); ws = torch.empty(2 * B * H * S, device=dev)
for _ in range(3):
    ops.attention_bwd(q, k, v, o, lse, do, delta, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], B, H, S, S, hd, ws=ws)
torch.cuda.synchronize()
nblk = (S // 128) * H * B
r = delta.view(-1)[:2 * nblk].view(nblk, 2).cpu()
ticks, real = r[:, 0], r[:, 1]
print("dK/dV IL kernel: blocks %d; shader ticks per block mean %.0f -> per 64-query tile %.0f; block time mean %.1f us; effective clock %.0f MHz" % (
    nblk, ticks.mean(), ticks.mean() / (S // 64), (real / 100).mean(), (ticks / real * 100).mean()))
print("MFMA cycles per tile per SIMD (2 waves x 28 x 32) = 1792 -> MFMA-busy in shader cycles %.1f %%" % (1792 / (ticks.mean() / (S // 64)) * 100))
