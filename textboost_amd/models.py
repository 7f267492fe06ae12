"""Shape specifications of the SD1.x/SD2.x UNet and CLIP text model state dicts (diffusers / transformers key names)
and seeded random initialisers with those shapes.

There is no network (and no checkpoint) on the build or GPU boxes, so benchmarks and parity tests run on
random-init weights of the exact architecture; a real `diffusion_pytorch_model.safetensors` /
`text_encoder/model.safetensors` state dict drops into HipUNet / HipTextEncoder unchanged (same keys).
Reference: the models the reference loads at train_textboost.py:646-656."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .text_encoder import CLIPGeometry, HF_LAYER
from .unet import UNetGeometry

SD15_UNET = UNetGeometry()
SD21_UNET = UNetGeometry(num_heads=(5, 10, 20, 20), cross_attention_dim=1024, use_linear_projection=True)
SD15_CLIP = CLIPGeometry()
# SD2.x text encoder: OpenCLIP ViT-H/14 text tower, 23 of its 24 layers (penultimate-layer output), erf-GELU MLP
SD21_CLIP = CLIPGeometry(hidden_size=1024, intermediate_size=4096, num_layers=23, num_heads=16, act="gelu")


def unet_shapes(geo: UNetGeometry) -> Dict[str, Tuple[int, ...]]:
    ch = geo.block_out_channels
    te = ch[0] * 4
    S: Dict[str, Tuple[int, ...]] = {}

    def wb(name, *shape):
        S[name + ".weight"] = tuple(shape)
        S[name + ".bias"] = (shape[0],)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); wb(p + ".conv1", cout, cin, 3, 3); wb(p + ".time_emb_proj", cout, te)
        norm(p + ".norm2", cout); wb(p + ".conv2", cout, cout, 3, 3)
        if cin != cout:
            wb(p + ".conv_shortcut", cout, cin, 1, 1)

    def transformer(p, c):
        norm(p + ".norm", c)
        if geo.use_linear_projection:
            wb(p + ".proj_in", c, c); wb(p + ".proj_out", c, c)
        else:
            wb(p + ".proj_in", c, c, 1, 1); wb(p + ".proj_out", c, c, 1, 1)
        tb = p + ".transformer_blocks.0"
        for n in ("norm1", "norm2", "norm3"):
            norm(f"{tb}.{n}", c)
        for x in "qkv":
            S[f"{tb}.attn1.to_{x}.weight"] = (c, c)
        wb(f"{tb}.attn1.to_out.0", c, c)
        S[f"{tb}.attn2.to_q.weight"] = (c, c)
        S[f"{tb}.attn2.to_k.weight"] = (c, geo.cross_attention_dim)
        S[f"{tb}.attn2.to_v.weight"] = (c, geo.cross_attention_dim)
        wb(f"{tb}.attn2.to_out.0", c, c)
        wb(f"{tb}.ff.net.0.proj", 8 * c, c)
        wb(f"{tb}.ff.net.2", c, 4 * c)

    wb("conv_in", ch[0], geo.in_channels, 3, 3)
    wb("time_embedding.linear_1", te, ch[0]); wb("time_embedding.linear_2", te, te)
    L_ = geo.layers_per_block
    prev = ch[0]
    skips = [ch[0]]
    for i, c in enumerate(ch):
        for j in range(L_):
            resnet(f"down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
            if geo.cross_attn_levels[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", c)
            skips.append(c)
        if i < len(ch) - 1:
            wb(f"down_blocks.{i}.downsamplers.0.conv", c, c, 3, 3)
            skips.append(c)
        prev = c
    resnet("mid_block.resnets.0", ch[-1], ch[-1]); transformer("mid_block.attentions.0", ch[-1]); resnet("mid_block.resnets.1", ch[-1], ch[-1])
    prev = ch[-1]
    for i, c in enumerate(reversed(ch)):
        lvl = len(ch) - 1 - i
        for j in range(L_ + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else c) + skips.pop(), c)
            if geo.cross_attn_levels[lvl]:
                transformer(f"up_blocks.{i}.attentions.{j}", c)
        if i < len(ch) - 1:
            wb(f"up_blocks.{i}.upsamplers.0.conv", c, c, 3, 3)
        prev = c
    norm("conv_norm_out", ch[0]); wb("conv_out", geo.out_channels, ch[0], 3, 3)
    return S


def clip_shapes(geo: CLIPGeometry) -> Dict[str, Tuple[int, ...]]:
    D, I = geo.hidden_size, geo.intermediate_size
    S = {"text_model.embeddings.token_embedding.weight": (geo.vocab_size, D),
         "text_model.embeddings.position_embedding.weight": (geo.max_pos, D),
         "text_model.final_layer_norm.weight": (D,), "text_model.final_layer_norm.bias": (D,)}
    for i in range(geo.num_layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("ln1", "ln2"):
            S[p + HF_LAYER[n] + ".weight"] = (D,); S[p + HF_LAYER[n] + ".bias"] = (D,)
        for n in ("q", "k", "v", "out"):
            S[p + HF_LAYER[n] + ".weight"] = (D, D); S[p + HF_LAYER[n] + ".bias"] = (D,)
        S[p + HF_LAYER["fc1"] + ".weight"] = (I, D); S[p + HF_LAYER["fc1"] + ".bias"] = (I,)
        S[p + HF_LAYER["fc2"] + ".weight"] = (D, I); S[p + HF_LAYER["fc2"] + ".bias"] = (D,)
    return S


def random_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, device="cpu", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """torch-default-like init: weights/biases U(-1/sqrt(fan_in), 1/sqrt(fan_in)); norm weights 1 (+small noise), biases 0;
    embeddings N(0, 0.02). Deterministic in (seed, device type)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    fan = {}
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) >= 2:
            f = 1
            for s in shp[1:]:
                f *= s
            fan[k[:-7]] = f
    for k, shp in shapes.items():
        base = k.rsplit(".", 1)[0]
        if "embedding.weight" in k:
            t = torch.randn(shp, generator=g, device=device, dtype=dtype) * 0.02
        elif "norm" in k.lower():
            t = torch.ones(shp, device=device, dtype=dtype) if k.endswith("weight") else torch.zeros(shp, device=device, dtype=dtype)
            t = t + 0.02 * torch.randn(shp, generator=g, device=device, dtype=dtype)
        else:
            b = fan.get(base, shp[0]) ** -0.5
            t = (torch.rand(shp, generator=g, device=device, dtype=dtype) * 2 - 1) * b
        sd[k] = t
    return sd


def count_params(shapes) -> int:
    n = 0
    for shp in shapes.values():
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def unet_geometry_from_config(cfg: dict) -> UNetGeometry:
    """`UNet2DConditionModel.from_pretrained(..., subfolder="unet")` (train_textboost.py:651-656): the geometry a diffusers `unet/config.json`
    describes.  `attention_head_dim` is the number of heads in SD1.x / SD2.x configs (8, or [5, 10, 20, 20]); anything this path does not
    implement (other block types, dual cross-attention, class embeddings, ...) is rejected instead of being silently ignored."""
    down = list(cfg.get("down_block_types", ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]))
    up = list(cfg.get("up_block_types", ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3))
    if set(down) - {"CrossAttnDownBlock2D", "DownBlock2D"} or set(up) - {"CrossAttnUpBlock2D", "UpBlock2D"}:
        raise NotImplementedError(f"UNet block types {down} / {up} are not built")
    cross = tuple(t == "CrossAttnDownBlock2D" for t in down)
    if tuple(t == "CrossAttnUpBlock2D" for t in reversed(up)) != cross:
        raise NotImplementedError("up_block_types must mirror down_block_types")
    for key, ok in (("class_embed_type", (None,)), ("addition_embed_type", (None,)), ("dual_cross_attention", (False, None)),
                    ("only_cross_attention", (False, None)), ("act_fn", ("silu", None)), ("mid_block_type", ("UNetMidBlock2DCrossAttn", None)),
                    ("transformer_layers_per_block", (1, None)), ("center_input_sample", (False, None)), ("flip_sin_to_cos", (True, None)),
                    ("freq_shift", (0, None)), ("downsample_padding", (1, None)), ("upcast_attention", (False, None))):
        if cfg.get(key) not in ok:
            raise NotImplementedError(f"unet config {key}={cfg.get(key)!r} is not built")
    heads = cfg.get("attention_head_dim", 8)
    heads = tuple(heads) if isinstance(heads, (list, tuple)) else int(heads)
    return UNetGeometry(in_channels=cfg.get("in_channels", 4), out_channels=cfg.get("out_channels", 4),
                        block_out_channels=tuple(cfg.get("block_out_channels", (320, 640, 1280, 1280))),
                        layers_per_block=cfg.get("layers_per_block", 2), cross_attn_levels=cross, num_heads=heads,
                        cross_attention_dim=cfg.get("cross_attention_dim", 768), norm_num_groups=cfg.get("norm_num_groups", 32),
                        norm_eps=cfg.get("norm_eps", 1e-5), use_linear_projection=bool(cfg.get("use_linear_projection", False)))


def clip_geometry_from_config(cfg: dict) -> CLIPGeometry:
    """transformers `CLIPTextConfig` (`text_encoder/config.json`) -> the text-encoder geometry (SD1.x CLIP-L, SD2.x OpenCLIP-H's 23 layers)."""
    act = cfg.get("hidden_act", "quick_gelu")
    if act not in ("quick_gelu", "gelu"):
        raise NotImplementedError(f"text encoder hidden_act={act!r} is not built")
    return CLIPGeometry(vocab_size=cfg.get("vocab_size", 49408), hidden_size=cfg.get("hidden_size", 768),
                        intermediate_size=cfg.get("intermediate_size", 3072), num_layers=cfg.get("num_hidden_layers", 12),
                        num_heads=cfg.get("num_attention_heads", 12), max_pos=cfg.get("max_position_embeddings", 77), act=act,
                        eps=cfg.get("layer_norm_eps", 1e-5))
