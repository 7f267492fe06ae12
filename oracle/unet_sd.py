"""ORACLE (test infrastructure only) -- eager-PyTorch fp32 restatement of the SD1.x / SD2.x
`UNet2DConditionModel` that the reference calls at train_textboost.py:1063-1067
(`unet(noisy, timesteps, encoder_hidden_states).sample`).

The arithmetic lives in diffusers==0.29.0 (pyproject.toml:12), which is NOT vendored under
/root/reference and is NOT installed in this image, so it is restated here from the published
layer specification.  Parameter names are identical to diffusers' state-dict
keys, so a real `diffusion_pytorch_model.safetensors` loads with `load_state_dict`.

Pinning: the reference has no tests or golden vectors for this path ("parity unpinned" for the
third-party arithmetic); what IS pinned (tests/test_oracle_*.py): the exact published parameter
count 859,520,964 for the SD1.x config, the timestep-embedding constants of SURVEY.md 8(c)4 and the
per-op equality of every block with torch's own CPU ops (conv2d/group_norm/sdpa/...).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # per level: does the down block (and the mirrored up block) carry transformer blocks?
    cross_attn_levels: Tuple[bool, ...] = (True, True, True, False)
    # diffusers calls this `attention_head_dim` but it is the NUMBER OF HEADS
    num_heads: Tuple[int, ...] | int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False  # SD2.x: True
    sample_size: int = 64

    def heads(self, level: int) -> int:
        return self.num_heads if isinstance(self.num_heads, int) else self.num_heads[level]

    @staticmethod
    def sd15() -> "UNetConfig":
        return UNetConfig()

    @staticmethod
    def sd21(sample_size: int = 64) -> "UNetConfig":
        return UNetConfig(num_heads=(5, 10, 20, 20), cross_attention_dim=1024,
                          use_linear_projection=True, sample_size=sample_size)

    @staticmethod
    def tiny(cross_dim: int = 64) -> "UNetConfig":
        """Small config for parity tests the CPU oracle finishes in seconds."""
        return UNetConfig(block_out_channels=(64, 128, 128, 128), num_heads=(2, 2, 4, 4),
                          cross_attention_dim=cross_dim, sample_size=16)


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`: cos half first."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    def __init__(self, dim, heads, cross_dim=None):
        super().__init__()
        self.heads = heads
        kv = cross_dim if cross_dim is not None else dim
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(kv, dim, bias=False)
        self.to_v = nn.Linear(kv, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.kv_lora_r = 0

    def add_kv_lora(self, r, alpha=None):
        """peft LoraConfig(r, lora_alpha=r, init_lora_weights="gaussian", target_modules=["attn2.to_k", "attn2.to_v"]) of
        /root/reference/train_textboost.py:712-721: y = W x + B (A x) * (alpha / r), A ~ N(0, (1/r)^2), B = 0."""
        kv, dim = self.to_k.in_features, self.to_k.out_features
        self.kv_lora_r, self.kv_lora_scaling = r, (alpha if alpha is not None else r) / r
        self.k_lora_A = nn.Parameter(torch.randn(r, kv) / r)
        self.k_lora_B = nn.Parameter(torch.zeros(dim, r))
        self.v_lora_A = nn.Parameter(torch.randn(r, kv) / r)
        self.v_lora_B = nn.Parameter(torch.zeros(dim, r))
        return [self.k_lora_A, self.k_lora_B, self.v_lora_A, self.v_lora_B]

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, S, C = x.shape
        hd = C // self.heads
        q = self.to_q(x).view(B, S, self.heads, hd).transpose(1, 2)
        k, v = self.to_k(ctx), self.to_v(ctx)
        if self.kv_lora_r:
            k = k + F.linear(F.linear(ctx, self.k_lora_A), self.k_lora_B) * self.kv_lora_scaling
            v = v + F.linear(F.linear(ctx, self.v_lora_A), self.v_lora_B) * self.kv_lora_scaling
        k = k.view(B, -1, self.heads, hd).transpose(1, 2)
        v = v.view(B, -1, self.heads, hd).transpose(1, 2)
        # AttnProcessor2_0: F.scaled_dot_product_attention, no mask, scale hd^-0.5
        p = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (p @ v).transpose(1, 2).reshape(B, S, C)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)  # erf gelu


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, cross_dim, groups, linear_proj):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        if linear_proj:
            self.proj_in = nn.Linear(dim, dim)
            self.proj_out = nn.Linear(dim, dim)
        else:
            self.proj_in = nn.Conv2d(dim, dim, 1)
            self.proj_out = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim)])

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x)
        if self.linear_proj:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        if self.linear_proj:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + res


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, level, cin, cout, temb_ch, add_down):
        super().__init__()
        L = cfg.layers_per_block
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, temb_ch, cfg.norm_num_groups, cfg.norm_eps) for j in range(L)])
        if cfg.cross_attn_levels[level]:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection) for _ in range(L)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for j, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, ch, temb_ch):
        super().__init__()
        lvl = len(cfg.block_out_channels) - 1
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, cfg.heads(lvl), cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cfg: UNetConfig, level, prev_out, cout, skip_chs: Sequence[int], temb_ch, add_up):
        super().__init__()
        L = cfg.layers_per_block + 1
        self.resnets = nn.ModuleList([
            ResnetBlock2D((prev_out if j == 0 else cout) + skip_chs[j], cout, temb_ch, cfg.norm_num_groups, cfg.norm_eps)
            for j in range(L)])
        if cfg.cross_attn_levels[level]:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, cfg.heads(level), cfg.cross_attention_dim, cfg.norm_num_groups, cfg.use_linear_projection) for _ in range(L)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for j, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[j](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DCondition(nn.Module):
    """`.forward(sample[B,4,h,w], timesteps[B], encoder_hidden_states[B,77,D]) -> eps[B,4,h,w]`."""

    def __init__(self, cfg: UNetConfig = UNetConfig()):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        temb_ch = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_ch)
        self.down_blocks = nn.ModuleList()
        skip_chs = [ch[0]]
        prev = ch[0]
        for i, c in enumerate(ch):
            last = i == len(ch) - 1
            self.down_blocks.append(DownBlock(cfg, i, prev, c, temb_ch, add_down=not last))
            skip_chs += [c] * cfg.layers_per_block + ([] if last else [c])
            prev = c
        self.mid_block = MidBlock(cfg, ch[-1], temb_ch)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        prev = ch[-1]
        for i, c in enumerate(rev):
            level = len(ch) - 1 - i
            sk = [skip_chs.pop() for _ in range(cfg.layers_per_block + 1)]
            self.up_blocks.append(UpBlock(cfg, level, prev, c, sk, temb_ch, add_up=i < len(ch) - 1))
            prev = c
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def add_crossattn_kv_adapters(self, r, alpha=None):
        """--unet_params_to_train crossattn_kv (/root/reference/train_textboost.py:712-721): rank-r adapters on every attn2.to_k / to_v.
        Returns {module path of the attn2: [k_A, k_B, v_A, v_B]} in module order (the third AdamW group, :838-841)."""
        out = {}
        for name, m in self.named_modules():
            if name.endswith(".attn2") and isinstance(m, Attention):
                out[name] = m.add_kv_lora(r, alpha)
        return out

    def forward(self, sample, timesteps, encoder_hidden_states):
        temb = self.time_embedding(timestep_embedding(timesteps, self.cfg.block_out_channels[0]).to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states)
        return self.conv_out(F.silu(self.conv_norm_out(x)))
