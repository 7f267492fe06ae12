"""ORACLE (test infrastructure only) -- eager-PyTorch fp32 restatement of the SD1.x / SD2.x VAE ENCODER path the
reference runs at the top of every training step, train_textboost.py:1036-1037:

    model_input = vae.encode(pixel_values).latent_dist.sample()
    model_input = model_input * vae.config.scaling_factor

(`vae` = diffusers `AutoencoderKL`, kept in fp32: `vae.to(accelerator.device, dtype=torch.float32)`, :938).
SURVEY.md 8(f) row 1 ("next" after the hot path).

The arithmetic lives in diffusers==0.29.0 (pyproject.toml:12), which is NOT vendored under /root/reference and NOT
installed here, so it is restated from the published architecture of `AutoencoderKL` (`Encoder`, `DownEncoderBlock2D`,
`UNetMidBlock2D` with one single-head `Attention`, `DiagonalGaussianDistribution`).  Parameter names are the diffusers
state-dict keys (>= 0.20 attention naming: `to_q/to_k/to_v/to_out.0`), so a real `vae/diffusion_pytorch_model.safetensors`
loads with `load_state_dict(strict=False)` (the decoder keys are ignored).

Pinning: "parity unpinned" for the third-party arithmetic (no reference tests, diffusers absent).  What IS pinned
(tests/test_oracle_vae.py): the published encoder parameter count 34,163,592 (+ 72 for `quant_conv`; encoder + decoder
49,490,179 + quant 72 + post_quant 20 = the well-known 83,653,863 of the SD VAE), and per-op equality of each block with
torch's own CPU ops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215  # SD1.x / SD2.x `vae.config.scaling_factor`

    @staticmethod
    def sd() -> "VAEConfig":
        return VAEConfig()

    @staticmethod
    def tiny() -> "VAEConfig":
        return VAEConfig(block_out_channels=(64, 64, 128, 128), layers_per_block=1)


class ResnetBlock2D(nn.Module):
    """diffusers ResnetBlock2D with temb_channels=None, output_scale_factor=1: x' + conv2(silu(gn2(conv1(silu(gn1 x)))))."""

    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Downsample2D(nn.Module):
    """diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then a stride-2 3x3 conv without padding."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class AttnBlock(nn.Module):
    """diffusers Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups, bias=True, upcast_softmax=True) as used
    by UNetMidBlock2D of the VAE: GroupNorm -> q,k,v Linear -> softmax(q k^T / sqrt(C)) v -> Linear -> + input."""

    def __init__(self, ch, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax((q @ k.transpose(1, 2)) * (C ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return x + o.transpose(1, 2).reshape(B, C, H, W)


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cfg: VAEConfig, cin, cout, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, cfg.norm_num_groups, cfg.norm_eps)
                                      for j in range(cfg.layers_per_block)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, cfg: VAEConfig, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([AttnBlock(ch, cfg.norm_num_groups, cfg.norm_eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.down_blocks.append(DownEncoderBlock2D(cfg, prev, c, add_down=i < len(ch) - 1))
            prev = c
        self.mid_block = MidBlock(cfg, ch[-1])
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAEEncoder(nn.Module):
    """`AutoencoderKL.encode(x).latent_dist` restricted to what the training step uses."""

    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def moments(self, pixel_values):
        """mean, logvar (clamped to [-30, 20] as DiagonalGaussianDistribution does) -- each [B, latent, H/8, W/8]."""
        m = self.quant_conv(self.encoder(pixel_values))
        mean, logvar = torch.chunk(m, 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)

    def encode_sample(self, pixel_values, noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        """train_textboost.py:1036-1037: `vae.encode(x).latent_dist.sample() * vae.config.scaling_factor`.
        sample = mean + exp(0.5 * logvar) * eps,  eps ~ N(0, 1) drawn with `randn_tensor` (here: `noise`, or torch.randn)."""
        mean, logvar = self.moments(pixel_values)
        if noise is None:
            noise = torch.randn(mean.shape, generator=generator, dtype=mean.dtype)
        return (mean + torch.exp(0.5 * logvar) * noise) * self.cfg.scaling_factor


def count_encoder_params(cfg: VAEConfig = VAEConfig()) -> Tuple[int, int]:
    """(encoder params, quant_conv params) without allocating the weights."""
    with torch.device("meta"):
        m = VAEEncoder(cfg)
    enc = sum(p.numel() for p in m.encoder.parameters())
    return enc, sum(p.numel() for p in m.quant_conv.parameters())
