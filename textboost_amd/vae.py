"""HIP executor of the SD VAE ENCODER -- the first thing every reference training step runs (train_textboost.py:1036-1037):

    model_input = vae.encode(pixel_values).latent_dist.sample() * vae.config.scaling_factor

SURVEY.md 8(f) row 1.  `vae` is diffusers' `AutoencoderKL` (state-dict keys `encoder.*`, `quant_conv.*`), frozen, forward only.
The reference keeps it in fp32 (:938); here it runs like the UNet -- fp16 operands / activations (NHWC), fp32 MFMA accumulation,
fp32 GroupNorm statistics, fp32 softmax, fp32 moments and sampling -- on the same kernels (`tb_gemm` implicit-GEMM 3x3 convs incl.
the LDS-halo kernel, `tb_groupnorm_fwd`), plus three small ones: `tb_convin_to_nhwc` (RGB conv_in), `tb_softmax_rows` and
`tb_vae_sample`.  Divergence from the reference: mixed precision instead of fp32 (latents agree with the fp32 CPU restatement to ~1e-3
relative; tolerance written in tests/test_gpu_vae.py).

  * Downsample2D(padding=0) = F.pad(x, (0,1,0,1)) + conv(stride 2): the conv gather with `shift = 1` (no padded copy).
  * The mid-block attention is ONE head of 512 channels over (H/8)*(W/8) pixels -- outside the flash kernels' head-dim range -- so it
    runs per image as GEMM (fp32 scores) -> row softmax -> GEMM.  V^T comes straight out of a GEMM with swapped operands
    (W_v h^T), and because softmax rows sum to one the value bias is added after the PV product, exactly.
  * `quant_conv` (1x1, 8 -> 8) is folded into `conv_out` on the host in fp32 (a composition of linear maps).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L
from . import ops
from .unet import _f32_via_f16, pack_conv3x3, pack_linear


@dataclass
class VAEGeometry:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


def vae_encoder_shapes(geo: VAEGeometry) -> Dict[str, Tuple[int, ...]]:
    """diffusers AutoencoderKL state-dict keys of the encoder half (+ quant_conv) -> shapes."""
    S: Dict[str, Tuple[int, ...]] = {}

    def wb(name, *shape):
        S[name + ".weight"] = tuple(shape)
        S[name + ".bias"] = (shape[0],)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); wb(p + ".conv1", cout, cin, 3, 3)
        norm(p + ".norm2", cout); wb(p + ".conv2", cout, cout, 3, 3)
        if cin != cout:
            wb(p + ".conv_shortcut", cout, cin, 1, 1)

    ch = geo.block_out_channels
    wb("encoder.conv_in", ch[0], geo.in_channels, 3, 3)
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(geo.layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < len(ch) - 1:
            wb(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3, 3)
        prev = c
    resnet("encoder.mid_block.resnets.0", prev, prev)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", prev)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        wb(f"{a}.{n}", prev, prev)
    resnet("encoder.mid_block.resnets.1", prev, prev)
    norm("encoder.conv_norm_out", prev)
    wb("encoder.conv_out", 2 * geo.latent_channels, prev, 3, 3)
    wb("quant_conv", 2 * geo.latent_channels, 2 * geo.latent_channels, 1, 1)
    return S


class HipVAEEncoder:
    """`encode(pixel_values[B,3,H,W] fp32 in [-1,1], noise=None) -> latents[B,4,H/8,W/8] fp32` (already times scaling_factor)."""

    def __init__(self, geo: VAEGeometry, state_dict: Dict[str, torch.Tensor], batch: int, height: int, width: int, device="cuda"):
        nd = len(geo.block_out_channels) - 1
        assert height % (1 << nd) == 0 and width % (1 << nd) == 0
        assert all(c % 64 == 0 for c in geo.block_out_channels)  # conv-as-GEMM needs Cin % 64 == 0
        self.geo, self.B, self.H, self.W, self.dev = geo, batch, height, width, device
        self.dtype = torch.float32  # what `vae.dtype` reports in the reference (:1027 casts pixel_values to it)
        self._bufs: Dict[str, torch.Tensor] = {}
        # The reference keeps the VAE in fp32 in EVERY --mixed_precision mode (train_textboost.py:938).  This encoder multiplies 16-bit operands with
        # fp32 accumulation / statistics; its 16-bit type is therefore pinned to IEEE half (11 significand bits; latents 1e-2 from an fp32 run at
        # full size) also when the process trains with the bfloat16 build (8 bits): it packs and runs on the fp16 library whatever `_lib.half_kind()` is.
        self.half = "fp16"
        with L.use_half(self.half):
            self._pack(state_dict)
        self.gn_ws = torch.empty((2048 + 2 * batch) * geo.norm_num_groups * 2, device=device, dtype=torch.float32)
        self.generator: Optional[torch.Generator] = None

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev, geo = self.dev, self.geo
        P: Dict[str, torch.Tensor] = {}
        self.P = P

        def conv(name):
            P[name + ".w"], _ = pack_conv3x3(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def lin(name):
            P[name + ".w"], _ = pack_linear(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def norm(name):
            P[name + ".g"] = _f32_via_f16(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def resnet(p, cin, cout):
            norm(p + ".norm1"); conv(p + ".conv1"); norm(p + ".norm2"); conv(p + ".conv2")
            if cin != cout:
                lin(p + ".conv_shortcut")

        ch = geo.block_out_channels
        w = _f32_via_f16(sd["encoder.conv_in.weight"], dev)  # [C0, 3, 3, 3] -> [(tap*Cin + ci), C0]
        P["conv_in.wp"] = w.permute(2, 3, 1, 0).reshape(9 * geo.in_channels, ch[0]).contiguous()
        P["conv_in.b"] = _f32_via_f16(sd["encoder.conv_in.bias"], dev)
        prev = ch[0]
        for i, c in enumerate(ch):
            for j in range(geo.layers_per_block):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
            if i < len(ch) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv")
            prev = c
        resnet("encoder.mid_block.resnets.0", prev, prev)
        a = "encoder.mid_block.attentions.0"
        norm(a + ".group_norm")
        # q and k projections as one GEMM; v is produced transposed (W_v h^T), its bias is added after the PV product
        wq, wk = sd[a + ".to_q.weight"], sd[a + ".to_k.weight"]
        P[a + ".qk.w"], _ = pack_linear(torch.cat([wq, wk], dim=0), dev)
        P[a + ".qk.b"] = _f32_via_f16(torch.cat([sd[a + ".to_q.bias"], sd[a + ".to_k.bias"]]), dev)
        P[a + ".v.w"], _ = pack_linear(sd[a + ".to_v.weight"], dev)
        P[a + ".v.b"] = _f32_via_f16(sd[a + ".to_v.bias"], dev)
        lin(a + ".to_out.0")
        resnet("encoder.mid_block.resnets.1", prev, prev)
        norm("encoder.conv_norm_out")
        # quant_conv o conv_out, composed in fp32:  W'[o] = sum_j Wq[o,j] Wout[j],  b' = Wq bout + bq
        wq_ = sd["quant_conv.weight"].detach().float().reshape(2 * geo.latent_channels, 2 * geo.latent_channels)
        wo = sd["encoder.conv_out.weight"].detach().float()
        wfold = torch.einsum("oj,jcyx->ocyx", wq_, wo)
        bfold = wq_ @ sd["encoder.conv_out.bias"].detach().float() + sd["quant_conv.bias"].detach().float()
        P["moments.w"], _ = pack_conv3x3(wfold, dev)
        P["moments.b"] = bfold.to(dev).contiguous()

    # ------------------------------------------------------------------ buffers
    def buf(self, name, rows, cols, dtype=None):
        dtype = L.half_dtype() if dtype is None else dtype
        t = self._bufs.get(name)
        if t is None or t.shape != (rows, cols) or t.dtype != dtype:
            t = torch.empty(rows, cols, device=self.dev, dtype=dtype)
            self._bufs[name] = t
        return t

    def _gn(self, x, name, y, HW, silu):
        C = x.shape[1]
        st = self.buf("gn.stats", self.B * self.geo.norm_num_groups, 2, torch.float32)
        ops.groupnorm_fwd(x, y, self.P[name + ".g"], self.P[name + ".b"], st, self.gn_ws, self.B, HW, C, self.geo.norm_num_groups,
                          self.geo.norm_eps, silu)

    def _conv(self, x, name, out, Hin, Win, Hout, Wout, stride=1, shift=0, upsample=0, **epi):
        geo = dict(B=self.B, Hin=Hin, Win=Win, Cin=x.shape[1], Hout=Hout, Wout=Wout, stride=stride, sign=1, upsample=upsample, transposed=0,
                   shift=shift)
        return ops.gemm(x, self.P[name + ".w"], out, conv=geo, bias=self.P[name + ".b"], **epi)

    def _resnet(self, p, x, H, W, tag):
        """x [M, Cin] -> [M, Cout]; buffers are recycled by role and size (forward only, nothing is saved)."""
        M, cin = x.shape
        cout = self.P[p + ".conv1.w"].shape[0]
        a = self.buf(f"a.{M}x{cin}", M, cin)
        self._gn(x, p + ".norm1", a, H * W, True)
        h = self.buf(f"h.{M}x{cout}", M, cout)
        self._conv(a, p + ".conv1", h, H, W, H, W)
        a2 = self.buf(f"a.{M}x{cout}", M, cout)
        self._gn(h, p + ".norm2", a2, H * W, True)
        if cin != cout:
            sc = self.buf(f"sc.{M}x{cout}", M, cout)
            ops.gemm(x, self.P[p + ".conv_shortcut.w"], sc, bias=self.P[p + ".conv_shortcut.b"])
            res = sc
        else:
            res = x
        out = self.buf(f"o{tag}.{M}x{cout}", M, cout)
        self._conv(a2, p + ".conv2", out, H, W, H, W, R=res)
        return out

    def _attention(self, x, H, W, a="encoder.mid_block.attentions.0"):
        B, HW = self.B, H * W
        M, C = x.shape
        P = self.P
        hn = self.buf(f"a.{M}x{C}", M, C)
        self._gn(x, a + ".group_norm", hn, HW, False)
        qk = self.buf("attn.qk", M, 2 * C)
        ops.gemm(hn, P[a + ".qk.w"], qk, bias=P[a + ".qk.b"])
        o = self.buf("attn.o", M, C)
        scores = self.buf("attn.s", HW, HW, torch.float32)
        probs = self.buf("attn.p", HW, HW)
        vt = self.buf("attn.vt", C, HW)
        for b in range(B):
            r = slice(b * HW, (b + 1) * HW)
            ops.gemm(qk[r, :C], qk[r, C:], scores, alpha=C ** -0.5)   # q_b k_b^T / sqrt(C), fp32
            ops.softmax_rows(scores, probs)
            ops.gemm(P[a + ".v.w"], hn[r], vt)                       # V_b^T = W_v h_b^T  [C, HW]
            ops.gemm(probs, vt, o[r], bias=P[a + ".v.b"])            # P V + b_v (rows of P sum to 1)
        out = self.buf(f"oA.{M}x{C}", M, C)
        ops.gemm(o, P[a + ".to_out.0.w"], out, bias=P[a + ".to_out.0.b"], R=x)
        return out

    # ------------------------------------------------------------------ forward
    def moments(self, pixel_values):
        """fp32 [B*h*w, 2L] NHWC rows = (mean | logvar) before the clamp."""
        with L.use_half(self.half):
            return self._moments(pixel_values)

    def _moments(self, pixel_values):
        geo, B, H, W = self.geo, self.B, self.H, self.W
        ch = geo.block_out_channels
        assert pixel_values.shape == (B, geo.in_channels, H, W) and pixel_values.dtype == torch.float32
        x = self.buf(f"oX.{B * H * W}x{ch[0]}", B * H * W, ch[0])
        ops.convin_to_nhwc(pixel_values.contiguous(), geo.in_channels, self.P["conv_in.wp"], self.P["conv_in.b"], x, B, H, W, ch[0])
        tag = 0
        for i, c in enumerate(ch):
            for j in range(geo.layers_per_block):
                x = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", x, H, W, "AB"[tag & 1])
                tag += 1
            if i < len(ch) - 1:
                Ho, Wo = H // 2, W // 2
                y = self.buf(f"oD.{B * Ho * Wo}x{c}", B * Ho * Wo, c)
                self._conv(x, f"encoder.down_blocks.{i}.downsamplers.0.conv", y, H, W, Ho, Wo, stride=2, shift=1)
                x, H, W = y, Ho, Wo
        x = self._resnet("encoder.mid_block.resnets.0", x, H, W, "C")
        x = self._attention(x, H, W)
        x = self._resnet("encoder.mid_block.resnets.1", x, H, W, "C")
        a = self.buf(f"a.{x.shape[0]}x{x.shape[1]}", x.shape[0], x.shape[1])
        self._gn(x, "encoder.conv_norm_out", a, H * W, True)
        mom = self.buf("moments", B * H * W, 2 * geo.latent_channels, torch.float32)
        self._conv(a, "moments", mom, H, W, H, W)
        return mom, H, W

    def encode(self, pixel_values, noise: Optional[torch.Tensor] = None):
        """`vae.encode(pixel_values).latent_dist.sample() * vae.config.scaling_factor` -> fp32 NCHW [B, L, h, w].
        noise: the eps ~ N(0,1) of DiagonalGaussianDistribution.sample (drawn with torch.randn on the device when None)."""
        geo = self.geo
        mom, h, w = self.moments(pixel_values)
        if noise is None:
            noise = torch.randn(self.B, geo.latent_channels, h, w, device=self.dev, generator=self.generator)
        lat = self.buf("latents", self.B * geo.latent_channels, h * w, torch.float32)
        with L.use_half(self.half):
            ops.vae_sample(mom, noise.contiguous(), lat, self.B, h * w, geo.latent_channels, geo.scaling_factor)
        return lat.view(self.B, geo.latent_channels, h, w)


class HipVAEDecoder(HipVAEEncoder):
    """`decode(latents[B,4,h,w] fp32) -> image[B,3,8h,8w] fp32 in [0,1]`: the tail of StableDiffusionPipeline.__call__ used by the reference's
    validation (`log_validation`, train_textboost.py:453-531) and inference.py -- `vae.decode(latents / scaling_factor).sample`, then
    `(image / 2 + 0.5).clamp(0, 1)`.  diffusers keys `post_quant_conv.*`, `decoder.*`.  Same kernels and precision policy as the
    encoder; Upsample2D (nearest x2 + 3x3 conv) is the conv gather with `upsample = 1` (no upsampled tensor is materialised)."""
    half = None   # the validation pipeline's VAE follows the run's weight_dtype (train_textboost.py:469-479 `torch_dtype=weight_dtype`): the active library

    def __init__(self, geo: VAEGeometry, state_dict: Dict[str, torch.Tensor], batch: int, latent_h: int, latent_w: int, device="cuda"):
        assert all(c % 64 == 0 for c in geo.block_out_channels)
        self.geo, self.B, self.h, self.w, self.dev = geo, batch, latent_h, latent_w, device
        self.H, self.W = latent_h << (len(geo.block_out_channels) - 1), latent_w << (len(geo.block_out_channels) - 1)
        self.dtype = torch.float32
        self._bufs: Dict[str, torch.Tensor] = {}
        self._pack_decoder(state_dict)
        self.gn_ws = torch.empty((2048 + 2 * batch) * geo.norm_num_groups * 2, device=device, dtype=torch.float32)

    def _pack_decoder(self, sd):
        dev, geo = self.dev, self.geo
        P: Dict[str, torch.Tensor] = {}
        self.P = P

        def conv(name):
            P[name + ".w"], _ = pack_conv3x3(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def lin(name):
            P[name + ".w"], _ = pack_linear(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def norm(name):
            P[name + ".g"] = _f32_via_f16(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32_via_f16(sd[name + ".bias"], dev)

        def resnet(p, cin, cout):
            norm(p + ".norm1"); conv(p + ".conv1"); norm(p + ".norm2"); conv(p + ".conv2")
            if cin != cout:
                lin(p + ".conv_shortcut")

        L4 = geo.latent_channels
        ch = list(reversed(geo.block_out_channels))
        P["pq.w"] = sd["post_quant_conv.weight"].detach().float().reshape(L4, L4).to(dev).contiguous()
        P["pq.b"] = sd["post_quant_conv.bias"].detach().float().to(dev).contiguous()
        w = _f32_via_f16(sd["decoder.conv_in.weight"], dev)  # [C0, 4, 3, 3] -> [(tap*4 + ci), C0]
        P["conv_in.wp"] = w.permute(2, 3, 1, 0).reshape(9 * L4, ch[0]).contiguous()
        P["conv_in.b"] = _f32_via_f16(sd["decoder.conv_in.bias"], dev)
        resnet("decoder.mid_block.resnets.0", ch[0], ch[0])
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm")
        P[a + ".qk.w"], _ = pack_linear(torch.cat([sd[a + ".to_q.weight"], sd[a + ".to_k.weight"]], dim=0), dev)
        P[a + ".qk.b"] = _f32_via_f16(torch.cat([sd[a + ".to_q.bias"], sd[a + ".to_k.bias"]]), dev)
        P[a + ".v.w"], _ = pack_linear(sd[a + ".to_v.weight"], dev)
        P[a + ".v.b"] = _f32_via_f16(sd[a + ".to_v.bias"], dev)
        lin(a + ".to_out.0")
        resnet("decoder.mid_block.resnets.1", ch[0], ch[0])
        prev = ch[0]
        for i, c in enumerate(ch):
            for j in range(geo.layers_per_block + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
            if i < len(ch) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv")
            prev = c
        norm("decoder.conv_norm_out")
        conv("decoder.conv_out")

    def decode(self, latents):
        geo, B, H, W = self.geo, self.B, self.h, self.w
        L4 = geo.latent_channels
        ch = list(reversed(geo.block_out_channels))
        assert latents.shape == (B, L4, H, W) and latents.dtype == torch.float32
        z = self.buf("z", B * L4, H * W, torch.float32)
        ops.chan_mix(latents.contiguous(), self.P["pq.w"], self.P["pq.b"], z, B, L4, H * W, scale=1.0 / geo.scaling_factor)
        x = self.buf(f"oX.{B * H * W}x{ch[0]}", B * H * W, ch[0])
        ops.convin_to_nhwc(z, L4, self.P["conv_in.wp"], self.P["conv_in.b"], x, B, H, W, ch[0])
        x = self._resnet("decoder.mid_block.resnets.0", x, H, W, "C")
        x = self._attention(x, H, W, a="decoder.mid_block.attentions.0")
        x = self._resnet("decoder.mid_block.resnets.1", x, H, W, "C")
        tag = 0
        for i, c in enumerate(ch):
            for j in range(geo.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, H, W, "AB"[tag & 1])
                tag += 1
            if i < len(ch) - 1:
                Ho, Wo = 2 * H, 2 * W
                y = self.buf(f"oU.{B * Ho * Wo}x{c}", B * Ho * Wo, c)
                self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", y, H, W, Ho, Wo, upsample=1)
                x, H, W = y, Ho, Wo
        a = self.buf(f"a.{x.shape[0]}x{x.shape[1]}", x.shape[0], x.shape[1])
        self._gn(x, "decoder.conv_norm_out", a, H * W, True)
        rgb = self.buf("rgb", B * H * W, 4, torch.float32)
        self._conv(a, "decoder.conv_out", rgb[:, :geo.in_channels], H, W, H, W)
        img = self.buf("image", B * geo.in_channels, H * W, torch.float32)
        ops.vae_image(rgb, img, B, H * W, geo.in_channels)
        return img.view(B, geo.in_channels, H, W)


def vae_decoder_shapes(geo: VAEGeometry) -> Dict[str, Tuple[int, ...]]:
    """diffusers AutoencoderKL state-dict keys of the decoder half (+ post_quant_conv) -> shapes."""
    S: Dict[str, Tuple[int, ...]] = {}

    def wb(name, *shape):
        S[name + ".weight"] = tuple(shape)
        S[name + ".bias"] = (shape[0],)

    def norm(name, c):
        S[name + ".weight"] = (c,)
        S[name + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); wb(p + ".conv1", cout, cin, 3, 3)
        norm(p + ".norm2", cout); wb(p + ".conv2", cout, cout, 3, 3)
        if cin != cout:
            wb(p + ".conv_shortcut", cout, cin, 1, 1)

    L4 = geo.latent_channels
    ch = list(reversed(geo.block_out_channels))
    wb("post_quant_conv", L4, L4, 1, 1)
    wb("decoder.conv_in", ch[0], L4, 3, 3)
    resnet("decoder.mid_block.resnets.0", ch[0], ch[0])
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", ch[0])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        wb(f"{a}.{n}", ch[0], ch[0])
    resnet("decoder.mid_block.resnets.1", ch[0], ch[0])
    prev = ch[0]
    for i, c in enumerate(ch):
        for j in range(geo.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i < len(ch) - 1:
            wb(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3, 3)
        prev = c
    norm("decoder.conv_norm_out", ch[-1])
    wb("decoder.conv_out", geo.in_channels, ch[-1], 3, 3)
    return S
