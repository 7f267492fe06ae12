"""HIP executor of the frozen SD1.x/SD2.x UNet2DConditionModel: forward and dgrad-only backward.

Mirrors the call the reference makes at train_textboost.py:1063-1067 --
`unet(noisy_model_input, timesteps, encoder_hidden_states).sample` -- and the part of
`accelerator.backward(loss)` (:1108) that flows through the frozen UNet into `encoder_hidden_states`
(the UNet has no trainable parameter by default, :696-698, so no weight gradients are formed and the backward
stops at the first cross-attention).

Everything numeric runs in the C-ABI kernels of libtextboost_hip.so (textboost_amd.ops); this file only owns
buffers, the layer schedule and the weight re-packing:
  * activations are NHWC fp16, viewed as [B*H*W, C] with explicit row strides, so the 12 skip tensors are written
    by their producers straight into the channel slice of the up-path concat buffer they will be read from
    (zero-copy torch.cat), and nearest-x2 upsampling is folded into the consuming conv's gather;
  * the 32 cross-attention K/V projections of `encoder_hidden_states` are hoisted into ONE GEMM (and one dgrad GEMM);
  * the 22 time_emb_proj Linears are one GEMM whose fp32 output is consumed as a per-sample row bias by conv1;
  * weights are frozen, so every dgrad operand (W^T, tap-transposed conv weights) is materialised once at load.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import ops


import os as _os
# GEGLU backward fused into the ff.net.2 dgrad GEMM's epilogue: with the lean wide-tile epilogue it saves the d(gated) round trip
# (A/B in one process: 35.23 -> 35.06 ms per step); through the 4-wave kernels' generic epilogue it was slower than the streaming kernel
FUSE_GEGLU_BWD = _os.environ.get("TB_FUSE_GEGLU_BWD", "1") == "1"
# the whole GEGLU feed-forward of the C = 320 blocks as one launch per direction (csrc/ff_fused.hip): the gated tensor / d(proj) never go to memory
FUSE_FF = _os.environ.get("TB_FUSE_FF", "1") == "1"
FUSE_FF_LN = _os.environ.get("TB_FUSE_FF_LN", "1") == "1"   # the fused feed-forward backward also applies norm3's LayerNorm backward (A/B switch)
# round 6: the fused feed-forward's launch also runs its row-local neighbours (ops.ff_fwd pre / post): bit 1 = attn2.to_out + residual + norm3 in
# front, bit 2 = proj_out + residual behind (A/B switch)
FF_CHAIN = int(_os.environ.get("TB_FF_CHAIN", "3"))
# ... and the two Linear -> LayerNorm -> Linear pairs of a 64x64-map block as one launch each (ops.chain320): bit 1 = proj_in -> norm1 -> qkv,
# bit 2 = attn1.to_out + residual -> norm2 -> attn2.to_q
ROW_CHAIN = int(_os.environ.get("TB_ROW_CHAIN", "3"))
MATERIALIZE_UPSAMPLE = _os.environ.get("TB_MATERIALIZE_UPSAMPLE", "1") == "1"  # A/B switch (see the up-block forward)
# LayerNorm fused into the neighbouring Linear's epilogue where a tile spans the row (C = 320, the 64x64 maps): forward into the producer of the
# residual stream, backward onto the accumulators of the dgrad GEMM that feeds it (round 3; 28 LayerNorm launches per step fewer)
FUSE_LN = _os.environ.get("TB_FUSE_LN", "1") == "1"
# round 5: where no tile spans the row (C = 640 / 1280), the block's three LayerNorms are FOLDED into the Linear behind each of them (gamma into the
# frozen weight, mean / rstd as per-row scalars of the consumer's epilogue, row statistics written by the producer of the residual stream):
# no LayerNorm launch and no normalised copy in the forward (A/B switch)
FOLD_LN = _os.environ.get("TB_FOLD_LN", "1") == "1"
# Upsampler convolutions (nearest x2 + conv3x3) as four 2x2-tap sub-pixel convolutions with pre-summed frozen weights (round 4): 2.25x fewer FLOP in
# the forward and in the dgrad, no materialised 4x map, no 2x2 gradient pooling pass.  The summed filter rows are rounded to fp16 once: a stated
# divergence from the 9-tap arithmetic, bounded in tests/test_gpu_gemm.py.  TB_SUBPIXEL=0 restores the materialised-upsample path (A/B).
SUBPIXEL = _os.environ.get("TB_SUBPIXEL", "1") == "1"
# the downsamplers' input gradient as a sub-pixel convolution of d out (ops.pack_strided_dgrad_subpixel); 0 = the 4-wave transposed gather
DOWN_DGRAD_SUBPIXEL = _os.environ.get("TB_DOWN_DGRAD_SUBPIXEL", "1") == "1"


@dataclass
class UNetGeometry:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attn_levels: Tuple[bool, ...] = (True, True, True, False)
    num_heads: Tuple[int, ...] | int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False

    def heads(self, level):
        return self.num_heads if isinstance(self.num_heads, int) else self.num_heads[level]


def _f16(t, dtype=None):
    return t.detach().to(L.half_dtype() if dtype is None else dtype)


def _f32_via_f16(t, dev, dtype=None):
    """unet.to(fp16) (:937) rounds every parameter to fp16; biases / norm affines are consumed as fp32 here.  In the fp32 (no-AMP) mode
    (`dtype` = float32: weight_dtype stays float32, :930-939) nothing is rounded."""
    return t.detach().to(L.half_dtype() if dtype is None else dtype).to(torch.float32).to(dev).contiguous()


def pack_conv3x3(w, dev, dtype=None):
    """[Co,Ci,3,3] -> fwd [Co, 9*Ci] (k = (ky*3+kx)*Ci + ci) and dgrad [Ci, 9*Co] (k = (ky*3+kx)*Co + co), fp16 (fp32 in the no-AMP mode)."""
    w = _f16(w, dtype).to(dev)
    fwd = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
    dg = w.permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous()
    return fwd, dg


def pack_linear(w, dev, dtype=None):
    w = _f16(w, dtype).to(dev)
    if w.dim() == 4:  # 1x1 conv
        w = w.reshape(w.shape[0], w.shape[1])
    return w.contiguous(), w.t().contiguous()


def pack_geglu_rows(w):
    """[2*inner, ...] (h rows then g rows) -> 32-row interleaved blocks [h0..31 | g0..31 | h32..63 | ...]."""
    inner = w.shape[0] // 2
    hs = w[:inner].reshape(inner // 32, 32, *w.shape[1:])
    gs = w[inner:].reshape(inner // 32, 32, *w.shape[1:])
    return torch.stack([hs, gs], dim=1).reshape(w.shape)


class HipUNet:
    """`forward(sample[B,4,h,w] fp16 NCHW, timesteps[B] i64, ehs[B*77, D] fp16) -> pred[B,4,h,w] fp16`,
    `backward(dpred[B,4,h,w] fp32) -> d_ehs[B*77, D] fp32` (gradient of the scaled loss).  With dtype = float32 (the reference's no-AMP
    mode) sample / ehs / pred are fp32 as well."""

    def __init__(self, geo: UNetGeometry, state_dict: Dict[str, torch.Tensor], batch: int, height: int, width: int,
                 text_len: int = 77, device="cuda", attn_fp8: bool = False, dtype=None):
        """dtype = torch.float16: the reference's --mixed_precision fp16 run (`unet.to(accelerator.device, dtype=weight_dtype)`, :937);
        torch.float32: its default no-AMP run (:298-308, :930-939) -- every weight, activation and gradient fp32 (csrc/f32_path.hip)."""
        dtype = L.half_dtype() if dtype is None else dtype   # (the active library's 16-bit float: fp16, or bf16 after _lib.set_half("bf16"))
        assert dtype in (L.half_dtype(), torch.float32)
        self.geo, self.B, self.H, self.W, self.T, self.dev = geo, batch, height, width, text_len, device
        self.dtype = dtype
        assert not (attn_fp8 and dtype != torch.float16)   # (the e4m3 P.V forward exists for the fp16 build only)
        # BASELINE.json configs[4]: e4m3 P.V in the forward of the hd = 40 self-attention layers (opt-in; fp16 everywhere else and in the backward)
        self.attn_fp8 = attn_fp8
        self._fp8_ws = None
        self._bufs: Dict[str, torch.Tensor] = {}
        self.tape: List = []
        self._pack(state_dict)
        # tb_groupnorm_ws_floats = B * chunks * G * 2 with B * chunks <= 2048 (+B when chunks clamps to 1)
        self.gn_ws = torch.empty((2048 + 2 * batch) * geo.norm_num_groups * 2, device=device, dtype=torch.float32)
        if dtype == torch.float32:  # the fp32 attention materialises its score matrices: reserve the largest one now (level 0 self-attention)
            ops.reserve_attention_f32(device, batch, geo.heads(0), height * width, max(height * width, text_len))

    # ------------------------------------------------------------------ buffers
    def buf(self, name, rows, cols, dtype=None):
        dtype = self.dtype if dtype is None else dtype
        key = name
        t = self._bufs.get(key)
        if t is None or t.shape != (rows, cols) or t.dtype != dtype:
            t = torch.empty(rows, cols, device=self.dev, dtype=dtype)
            self._bufs[key] = t
        return t

    def scratch(self, tag, rows, cols, dtype=None):
        dtype = self.dtype if dtype is None else dtype
        return self.buf(f"scr.{tag}.{rows}x{cols}.{dtype}", rows, cols, dtype)

    # ------------------------------------------------------------------ cross-attention K/V adapters
    def enable_kv_lora(self, r: int, alpha: Optional[float] = None, seed: Optional[int] = None):
        """--unet_params_to_train crossattn_kv (train_textboost.py:712-721): peft LoraConfig(r, lora_alpha=r, "gaussian",
        target_modules=["attn2.to_k", "attn2.to_v"]) on the otherwise frozen UNet; its parameters are the optimizer's third group
        (:838-841).  fp32 and bf16 modes -- under --mixed_precision fp16 the reference casts these parameters to fp16 (:937) and GradScaler
        refuses them, so that combination cannot run there either.  Layout: kv_lora_A [n_layers, 2r, Dc] (k rows, then v rows), kv_lora_B
        [kv_total, r] (row n = output feature n of the concatenated K/V projections, i.e. aligned with the columns of the hoisted kv_all GEMM).
        bf16 mode (round 6; the reference trains bf16 adapter PARAMETERS there, :937): fp32 masters like the text encoder's LoRA, the forward
        multiplies bf16 copies of A_all / W2 as the hoisted GEMM's operands (the adapter output is rounded to bf16 like peft's bf16 module), the
        parameter gradients are the fp32 path's products on fp32 copies of the bf16 operands."""
        assert self.dtype == torch.float32 or L.half_kind() == "bf16", "UNet adapters: fp32 (no-AMP) or bf16 mode (fp16: GradScaler refuses fp16 parameters)"
        dev, Dc, n = self.dev, self.geo.cross_attention_dim, len(self.xattn)
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        self.kv_r = r
        self.kv_scaling = (alpha if alpha is not None else r) / r
        self.kv_lora_A = (torch.randn(n, 2 * r, Dc, generator=g) / r).to(dev)
        self.kv_lora_B = torch.zeros(self.kv_total, r, device=dev)
        self.kv_grad_A = torch.zeros_like(self.kv_lora_A)
        self.kv_grad_B = torch.zeros_like(self.kv_lora_B)
        cb = torch.empty(self.kv_total, dtype=torch.int32)
        for l, (p, C) in enumerate(self.xattn):
            ko = self.kv_off[p]
            cb[ko:ko + C] = l * 2 * r
            cb[ko + C:ko + 2 * C] = l * 2 * r + r
        self.kv_col_base = cb.to(dev)
        self.kv_w2 = torch.zeros(self.kv_total, n * 2 * r, device=dev)
        if self.dtype != torch.float32:   # 16-bit operand copies, the K extension padded to whole 64-wide k-tiles (zero columns / rows)
            n2r = n * 2 * r
            self.kv_n2r_pad = (n2r + 63) // 64 * 64
            self.kv_A16 = torch.zeros(self.kv_n2r_pad, Dc, device=dev, dtype=self.dtype)
            self.kv_w2_16 = torch.zeros(self.kv_total, self.kv_n2r_pad, device=dev, dtype=self.dtype)

    def pack_kv_lora(self):
        """refresh the K-extension operand from the adapter masters (once per optimizer step, like HipTextEncoder.pack_lora)"""
        if getattr(self, "kv_r", 0):
            ops.kv_lora_pack(self.kv_lora_B, self.kv_col_base, self.kv_w2, self.kv_r, self.kv_scaling)
            if self.dtype != torch.float32:
                n2r = self.kv_w2.shape[1]
                ops.convert(self.kv_w2, self.kv_w2_16[:, :n2r])
                ops.convert(self.kv_lora_A.view(n2r, -1), self.kv_A16[:n2r])

    def merged_kv_weight(self):
        """attn2.to_k / to_v weights with the adapters folded in, fp32 [kv_total, Dc]: W + W2 A_all (W2 = the block-structured scaling * B of
        pack_kv_lora) -- what a UNet WITHOUT adapter support loads to compute the same projections.  The validation sampler
        (train_textboost.py:453-531 samples with the trained unet) takes it: tests/test_gpu_f32.py."""
        assert getattr(self, "kv_r", 0)
        self.pack_kv_lora()
        n2r, Dc = self.kv_w2.shape[1], self.geo.cross_attention_dim
        base = self.P["kv_all.w"].float()
        out = torch.empty_like(base)
        ops.gemm_f32_t(self.kv_w2, self.kv_lora_A.view(n2r, Dc), out, self.kv_total, Dc, n2r, w_trans=True, R=base)
        return out

    def load_kv_weight(self, W):
        """replace every attn2.to_k / to_v weight (fp32 [kv_total, Dc], the layout of merged_kv_weight) in this executor's operand dtype"""
        assert tuple(W.shape) == tuple(self.P["kv_all.w"].shape)
        self.P["kv_all.w"].copy_(W.to(self.P["kv_all.w"].dtype))

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev, geo, wdt = self.dev, self.geo, self.dtype
        P = {}
        self.P = P
        _f32v = lambda t, d: _f32_via_f16(t, d, wdt)  # noqa: E731

        def conv(name):
            P[name + ".w"], P[name + ".wd"] = pack_conv3x3(sd[name + ".weight"], dev, wdt)
            P[name + ".b"] = _f32v(sd[name + ".bias"], dev)

        def lin(name, bias=True):
            P[name + ".w"], P[name + ".wd"] = pack_linear(sd[name + ".weight"], dev, wdt)
            if bias:
                P[name + ".b"] = _f32v(sd[name + ".bias"], dev)

        def norm(name):
            P[name + ".g"] = _f32v(sd[name + ".weight"], dev)
            P[name + ".b"] = _f32v(sd[name + ".bias"], dev)

        ch = geo.block_out_channels
        self.resnets: List[str] = []
        self.xattn: List[Tuple[str, int]] = []   # (prefix of attn2, C)
        self._walk = []
        # conv_in / conv_out as boundary kernels (4-channel NCHW side)
        w = _f32v(sd["conv_in.weight"], dev)                     # [C0,4,3,3]
        P["conv_in.wp"] = w.permute(2, 3, 1, 0).reshape(36, ch[0]).contiguous()
        P["conv_in.b"] = _f32v(sd["conv_in.bias"], dev)
        w = _f32v(sd["conv_out.weight"], dev)                    # [4,C0,3,3]
        P["conv_out.wp"] = w.permute(0, 2, 3, 1).reshape(4, 9, ch[0]).contiguous()
        P["conv_out.wdp"] = w.permute(2, 3, 0, 1).reshape(36, ch[0]).contiguous()
        P["conv_out.b"] = _f32v(sd["conv_out.bias"], dev)
        norm("conv_norm_out")
        lin("time_embedding.linear_1")
        lin("time_embedding.linear_2")

        def resnet(prefix, cin, cout):
            norm(prefix + ".norm1"); conv(prefix + ".conv1"); norm(prefix + ".norm2"); conv(prefix + ".conv2")
            if cin != cout:
                lin(prefix + ".conv_shortcut")
            self.resnets.append(prefix)

        self.fold_slots: Dict[str, int] = {}   # transformer prefix -> statistic slots per row (> 0: its LayerNorms are folded, see FOLD_LN)

        def transformer(prefix, C, level):
            norm(prefix + ".norm")
            lin(prefix + ".proj_in"); lin(prefix + ".proj_out")
            tb = prefix + ".transformer_blocks.0"
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{tb}.{n}")
            wq = torch.cat([sd[f"{tb}.attn1.to_{x}.weight"] for x in "qkv"], dim=0)
            lin(f"{tb}.attn1.to_out.0")
            lin(f"{tb}.attn2.to_out.0")
            self.xattn.append((f"{tb}.attn2", C))
            wff = pack_geglu_rows(sd[f"{tb}.ff.net.0.proj.weight"])
            bff = pack_geglu_rows(sd[f"{tb}.ff.net.0.proj.bias"])
            lin(f"{tb}.ff.net.2")
            M = self.B * (self.H >> level) * (self.W >> level)
            slots = 0
            if FOLD_LN and wdt != torch.float32 and not (FUSE_LN and ops.gemm_ln_ok(M, C, C, wdt)):
                slots = ops.lnfold_slots(M, C, wdt)
            self.fold_slots[prefix] = slots
            if slots:
                # W' = gamma (.) W in the weight's place (forward and dgrad: d(LN input) needs gamma (.) dy = dY W'), c1 / c2 beside it
                for key, w_, b_, ln in ((f"{tb}.attn1.qkv", wq, None, "norm1"), (f"{tb}.attn2.to_q", sd[f"{tb}.attn2.to_q.weight"], None, "norm2"),
                                        (f"{tb}.ff1", wff, bff, "norm3")):
                    wp, c1, c2 = ops.fold_layernorm(w_.to(dev), sd[f"{tb}.{ln}.weight"].to(dev), sd[f"{tb}.{ln}.bias"].to(dev),
                                                    None if b_ is None else b_.to(dev), wdt)
                    P[key + ".w"], P[key + ".wd"] = wp, wp.t().contiguous()
                    P[key + ".c1"], P[key + ".c2"] = c1, c2
                if "ones" not in P or P["ones"].numel() < C:
                    P["ones"] = torch.ones(C, device=dev, dtype=torch.float32)
                return
            P[f"{tb}.attn1.qkv.w"], P[f"{tb}.attn1.qkv.wd"] = pack_linear(wq, dev, wdt)
            lin(f"{tb}.attn2.to_q", bias=False)
            P[f"{tb}.ff1.w"], P[f"{tb}.ff1.wd"] = pack_linear(wff, dev, wdt)
            P[f"{tb}.ff1.b"] = _f32v(bff, dev)

        L_ = geo.layers_per_block
        prev = ch[0]
        skip_chs = [ch[0]]
        for i, c in enumerate(ch):
            for j in range(L_):
                resnet(f"down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
                if geo.cross_attn_levels[i]:
                    transformer(f"down_blocks.{i}.attentions.{j}", c, i)
                skip_chs.append(c)
            if i < len(ch) - 1:
                conv(f"down_blocks.{i}.downsamplers.0.conv")
                name = f"down_blocks.{i}.downsamplers.0.conv"
                # the stride-2 convolution's input gradient as a sub-pixel convolution over d out (csrc/gemm8.hip SUB = 1) where the coarse map tiles
                if DOWN_DGRAD_SUBPIXEL and wdt != torch.float32 and ops.subpixel_ok(self.B, self.H >> (i + 1), self.W >> (i + 1), c, c, wdt):
                    P[name + ".wdsub2"] = ops.pack_strided_dgrad_subpixel(sd[name + ".weight"].to(dev).to(wdt))
                skip_chs.append(c)
            prev = c
        resnet("mid_block.resnets.0", ch[-1], ch[-1])
        transformer("mid_block.attentions.0", ch[-1], len(ch) - 1)
        resnet("mid_block.resnets.1", ch[-1], ch[-1])
        self.skip_chs = list(skip_chs)
        rev = list(reversed(ch))
        prev = ch[-1]
        sk = list(skip_chs)
        for i, c in enumerate(rev):
            level = len(ch) - 1 - i
            for j in range(L_ + 1):
                s = sk.pop()
                resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else c) + s, c)
                if geo.cross_attn_levels[level]:
                    transformer(f"up_blocks.{i}.attentions.{j}", c, level)
            if i < len(ch) - 1:
                conv(f"up_blocks.{i}.upsamplers.0.conv")
                name = f"up_blocks.{i}.upsamplers.0.conv"
                if wdt != torch.float32 and ops.subpixel_ok(self.B, self.H >> level, self.W >> level, c, c, wdt):
                    P[name + ".wsub"], P[name + ".wdsub"] = ops.pack_subpixel_weights(sd[name + ".weight"].to(dev).to(wdt))
            prev = c
        # hoisted projections
        tw = torch.cat([sd[p + ".time_emb_proj.weight"] for p in self.resnets], dim=0)
        P["temb_all.w"] = _f16(tw, wdt).to(dev).contiguous()
        P["temb_all.b"] = _f32v(torch.cat([sd[p + ".time_emb_proj.bias"] for p in self.resnets], dim=0), dev)
        self.temb_off, o = {}, 0
        for p in self.resnets:
            self.temb_off[p] = o
            o += sd[p + ".time_emb_proj.weight"].shape[0]
        self.temb_total = o
        kv = torch.cat([torch.cat([sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], dim=0) for p, _ in self.xattn], dim=0)
        P["kv_all.w"], P["kv_all.wd"] = pack_linear(kv, dev, wdt)
        self.kv_off, o = {}, 0
        for p, C in self.xattn:
            self.kv_off[p] = o
            o += 2 * C
        self.kv_total = o
        self.first_xattn = self.xattn[0][0]

    # ------------------------------------------------------------------ primitive wrappers
    def _conv(self, x, name, out, B, Hin, Win, Hout, Wout, dgrad=False, stride=1, upsample=0, transposed=0, **epi):
        """returns `out`, or (with defer=True, when the launch split K) the ops.SplitKPartials the following GroupNorm consumes"""
        w = self.P[name + (".wd" if dgrad else ".w")]
        cin = x.shape[1]
        geo = dict(B=B, Hin=Hin, Win=Win, Cin=cin, Hout=Hout, Wout=Wout, stride=stride, sign=-1 if (dgrad and not transposed) else 1,
                   upsample=upsample, transposed=transposed)
        return ops.gemm(x, w, out, conv=geo, **epi)

    def _gn_fwd(self, x, name, y, stats, HW, silu, eps=None, partials=None):
        C = x.shape[1]
        ops.groupnorm_fwd(x, y, self.P[name + ".g"], self.P[name + ".b"], stats, self.gn_ws, self.B, HW, C, self.geo.norm_num_groups,
                          self.geo.norm_eps if eps is None else eps, silu, partials=partials)

    def _gn_bwd(self, dy, x, name, stats, dx, HW, silu, add=None, partials=None):
        C = x.shape[1]
        ops.groupnorm_bwd(dy, x, self.P[name + ".g"], self.P[name + ".b"], stats, dx, self.gn_ws, self.B, HW, C,
                          self.geo.norm_num_groups, silu, add, partials=partials)

    def _gn_splitk(self, HW, C):
        """may the producer of a GroupNorm input over [B*HW, C] leave its split-K slices to the GroupNorm launch (round 4: on the 16x16 / 8x8 maps
        the reducer launch between a convolution and the GroupNorm behind it was ~9 us of a ~45 us pair)?"""
        return ops.groupnorm_splitk_ok(self.B, HW, C, self.geo.norm_num_groups, self.dtype)

    @staticmethod
    def _pk(r):
        return r if isinstance(r, ops.SplitKPartials) else None

    # ------------------------------------------------------------------ blocks
    def _resnet(self, prefix, x, out, lvl_hw, x_partials=None, defer_out=False):
        """x [M,Cin] view -> out [M,Cout] view; returns its backward.
        x_partials: x has not been written yet -- its producer left split-K slices for norm1 to add (and to write x).
        defer_out: the caller guarantees that the NEXT launch on the stream is a GroupNorm over `out` alone; conv2 may then leave its
        split-K slices to it.  Returns (bwd, partials-or-None)."""
        H, W = lvl_hw
        HW, M, B = H * W, x.shape[0], self.B
        cin, cout = x.shape[1], out.shape[1]
        P = self.P
        st1 = self.buf(prefix + ".st1", B * self.geo.norm_num_groups, 2, torch.float32)
        st2 = self.buf(prefix + ".st2", B * self.geo.norm_num_groups, 2, torch.float32)
        a1 = self.scratch("a", M, cin)
        self._gn_fwd(x, prefix + ".norm1", a1, st1, HW, True, partials=x_partials)
        h1 = self.buf(prefix + ".h1", M, cout)
        rb = self.rowbias[:, self.temb_off[prefix]: self.temb_off[prefix] + cout]
        sk_out = self._gn_splitk(HW, cout)
        pk1 = self._pk(self._conv(a1, prefix + ".conv1", h1, B, H, W, H, W, bias=P[prefix + ".conv1.b"], rowbias=rb, rows_per_group=HW,
                                  defer=sk_out))
        a2 = self.scratch("a", M, cout)
        self._gn_fwd(h1, prefix + ".norm2", a2, st2, HW, True, partials=pk1)
        has_sc = cin != cout
        defer_out = defer_out and sk_out
        if has_sc:
            ops.gemm(x, P[prefix + ".conv_shortcut.w"], out, bias=P[prefix + ".conv_shortcut.b"])
            pk_out = self._pk(self._conv(a2, prefix + ".conv2", out, B, H, W, H, W, bias=P[prefix + ".conv2.b"], R=out, defer=defer_out))
        else:
            pk_out = self._pk(self._conv(a2, prefix + ".conv2", out, B, H, W, H, W, bias=P[prefix + ".conv2.b"], R=x, defer=defer_out))
        sk_in = self._gn_splitk(HW, cin)

        def bwd(dout, dx):
            da2 = self.scratch("g1", M, cout)
            pk = self._pk(self._conv(dout, prefix + ".conv2", da2, B, H, W, H, W, dgrad=True, defer=sk_out))
            dh1 = self.scratch("g2", M, cout)
            self._gn_bwd(da2, h1, prefix + ".norm2", st2, dh1, HW, True, partials=pk)
            da1 = self.scratch("g1", M, cin)
            if has_sc:   # (the shortcut's dgrad first: the GroupNorm backward must directly follow the convolution whose slices it adds)
                dsc = self.scratch("g3", M, cin)
                ops.gemm(dout, P[prefix + ".conv_shortcut.wd"], dsc)
                pk = self._pk(self._conv(dh1, prefix + ".conv1", da1, B, H, W, H, W, dgrad=True, defer=sk_in))
                self._gn_bwd(da1, x, prefix + ".norm1", st1, dx, HW, True, add=dsc, partials=pk)
            else:
                pk = self._pk(self._conv(dh1, prefix + ".conv1", da1, B, H, W, H, W, dgrad=True, defer=sk_in))
                self._gn_bwd(da1, x, prefix + ".norm1", st1, dx, HW, True, add=dout, partials=pk)
        return bwd, pk_out

    def _transformer(self, prefix, x, out, lvl_hw, level, x_partials=None):
        H, W = lvl_hw
        HW, M, B, C = H * W, x.shape[0], self.B, x.shape[1]
        heads = self.geo.heads(level)
        hd = C // heads
        P, T = self.P, self.T
        tb = prefix + ".transformer_blocks.0"
        G = self.geo.norm_num_groups
        st0 = self.buf(prefix + ".st0", B * G, 2, torch.float32)
        n0 = self.scratch("a", M, C)
        self._gn_fwd(x, prefix + ".norm", n0, st0, HW, False, eps=1e-6, partials=x_partials)  # (x_partials: see _resnet)
        t0 = self.buf(prefix + ".t0", M, C)
        fuse_ln = FUSE_LN and ops.gemm_ln_ok(M, C, C, self.dtype)
        # folded LayerNorms (FOLD_LN, decided in _pack): the producers of the residual stream (proj_in, attn1.to_out, attn2.to_out) write per-tile row
        # statistics (rs), the Linear behind each LayerNorm multiplies the RAW stream by gamma-folded weights and normalises in its epilogue
        fold = self.fold_slots.get(prefix, 0)
        assert not (fold and fuse_ln)
        rs = self.scratch("rs", M, 16 * 2, torch.float32).view(M, 16, 2) if fold else None
        ln_g = (lambda n: P["ones"]) if fold else (lambda n: P[f"{tb}.{n}.g"])    # backward: the dgrad through W' already carries gamma
        # --- self attention
        ls1 = self.buf(prefix + ".ls1", M, 2, torch.float32)
        qkv = self.buf(prefix + ".qkv", M, 3 * C)
        if fold:
            ops.gemm(n0, P[prefix + ".proj_in.w"], t0, bias=P[prefix + ".proj_in.b"], rs_out=rs, rs_slots=fold)
            ops.gemm(t0, P[tb + ".attn1.qkv.w"], qkv, bias=P[tb + ".attn1.qkv.c2"], lnfold=(rs, fold, P[tb + ".attn1.qkv.c1"], ls1, 1e-5))
        else:
            # round 6 (ROW_CHAIN): proj_in -> norm1 -> qkv as ONE launch on the 64x64 maps (csrc/chain320.hip): a 128-row tile spans the 320-wide rows of
            # all three layers, LayerNorm(t0) never goes to memory
            chain_a = ROW_CHAIN & 1 and fuse_ln and ops.chain320_ok(M, 3 * C, self.dtype)
            if chain_a:
                ops.chain320(n0, P[prefix + ".proj_in.w"], P[prefix + ".proj_in.b"], None, t0, P[tb + ".norm1.g"], P[tb + ".norm1.b"], ls1,
                             P[tb + ".attn1.qkv.w"], None, qkv)
            else:
                l1 = self.scratch("a2" if fuse_ln else "a", M, C)   # (fused: written while n0 -- scratch "a" -- is still being read)
                if fuse_ln:
                    ops.gemm(n0, P[prefix + ".proj_in.w"], t0, bias=P[prefix + ".proj_in.b"],
                             ln_fwd=(P[tb + ".norm1.g"], P[tb + ".norm1.b"], ls1, l1, 1e-5))
                else:
                    ops.gemm(n0, P[prefix + ".proj_in.w"], t0, bias=P[prefix + ".proj_in.b"])
                    ops.layernorm_fwd(t0, l1, P[tb + ".norm1.g"], P[tb + ".norm1.b"], ls1)
                ops.gemm(l1, P[tb + ".attn1.qkv.w"], qkv)
        o1 = self.buf(prefix + ".o1", M, C)
        lse1 = self.buf(prefix + ".lse1", B * heads, HW, torch.float32)
        fp8_ws = None
        if self.attn_fp8 and hd == 40 and HW % 256 == 0:
            if self._fp8_ws is None:
                self._fp8_ws = ops.attention_fp8_workspace(B, heads, HW, self.dev)  # one scratch, reused by every such layer (stream order)
            fp8_ws = self._fp8_ws
        ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o1, lse1, B, heads, HW, HW, hd, fp8_ws=fp8_ws)
        t1 = self.buf(prefix + ".t1", M, C)
        # --- cross attention (K/V hoisted)
        ls2 = self.buf(prefix + ".ls2", M, 2, torch.float32)
        q2 = self.buf(prefix + ".q2", M, C)
        if fold:
            ops.gemm(o1, P[tb + ".attn1.to_out.0.w"], t1, bias=P[tb + ".attn1.to_out.0.b"], R=t0, rs_out=rs, rs_slots=fold)
            ops.gemm(t1, P[tb + ".attn2.to_q.w"], q2, bias=P[tb + ".attn2.to_q.c2"], lnfold=(rs, fold, P[tb + ".attn2.to_q.c1"], ls2, 1e-5))
        else:
            chain_b = ROW_CHAIN & 2 and fuse_ln and ops.chain320_ok(M, C, self.dtype)   # attn1.to_out + residual -> norm2 -> attn2.to_q, one launch
            if chain_b:
                ops.chain320(o1, P[tb + ".attn1.to_out.0.w"], P[tb + ".attn1.to_out.0.b"], t0, t1, P[tb + ".norm2.g"], P[tb + ".norm2.b"], ls2,
                             P[tb + ".attn2.to_q.w"], None, q2)
            else:
                l2 = self.scratch("a", M, C)
                if fuse_ln:
                    ops.gemm(o1, P[tb + ".attn1.to_out.0.w"], t1, bias=P[tb + ".attn1.to_out.0.b"], R=t0,
                             ln_fwd=(P[tb + ".norm2.g"], P[tb + ".norm2.b"], ls2, l2, 1e-5))
                else:
                    ops.gemm(o1, P[tb + ".attn1.to_out.0.w"], t1, bias=P[tb + ".attn1.to_out.0.b"], R=t0)
                    ops.layernorm_fwd(t1, l2, P[tb + ".norm2.g"], P[tb + ".norm2.b"], ls2)
                ops.gemm(l2, P[tb + ".attn2.to_q.w"], q2)
        self._ensure_kv()
        ko = self.kv_off[tb + ".attn2"]
        k2, v2 = self.kv_all[:, ko:ko + C], self.kv_all[:, ko + C:ko + 2 * C]
        o2 = self.buf(prefix + ".o2", M, C)
        lse2 = self.buf(prefix + ".lse2", B * heads, HW, torch.float32)
        ops.attention_fwd(q2, k2, v2, o2, lse2, B, heads, HW, T, hd)
        t2 = self.buf(prefix + ".t2", M, C)
        # --- GEGLU feed-forward
        ls3 = self.buf(prefix + ".ls3", M, 2, torch.float32)
        l3 = None
        fuse_ff = FUSE_FF and not fold and ops.ff_fused_ok(M, C, 4 * C, self.dtype)
        # round 6 (FF_CHAIN bits: 1 = attn2.to_out + residual + norm3 in FRONT of the fused feed-forward, 2 = proj_out + residual BEHIND it, in the same
        # launch): a 128-row tile spans the 320-wide rows of all four layers, so l3 and t3 never leave the CU
        chain = FF_CHAIN if (fuse_ff and fuse_ln and self.dtype != torch.float32) else 0
        if fold:
            ops.gemm(o2, P[tb + ".attn2.to_out.0.w"], t2, bias=P[tb + ".attn2.to_out.0.b"], R=t1, rs_out=rs, rs_slots=fold)
        elif chain & 1:
            pass   # (issued with the feed-forward below)
        else:
            l3 = self.scratch("a", M, C)
            if fuse_ln:
                ops.gemm(o2, P[tb + ".attn2.to_out.0.w"], t2, bias=P[tb + ".attn2.to_out.0.b"], R=t1,
                         ln_fwd=(P[tb + ".norm3.g"], P[tb + ".norm3.b"], ls3, l3, 1e-5))
            else:
                ops.gemm(o2, P[tb + ".attn2.to_out.0.w"], t2, bias=P[tb + ".attn2.to_out.0.b"], R=t1)
                ops.layernorm_fwd(t2, l3, P[tb + ".norm3.g"], P[tb + ".norm3.b"], ls3)
        raw = self.buf(prefix + ".raw", M, 8 * C)
        if chain:
            pre = (P[tb + ".attn2.to_out.0.w"], P[tb + ".attn2.to_out.0.b"], t1, t2, P[tb + ".norm3.g"], P[tb + ".norm3.b"], ls3, 1e-5) if chain & 1 else None
            post = (P[prefix + ".proj_out.w"], P[prefix + ".proj_out.b"], x, out) if chain & 2 else None
            t3 = None if chain & 2 else self.scratch("a2", M, C)
            ops.ff_fwd(o2 if chain & 1 else l3, P[tb + ".ff1.w"], P[tb + ".ff1.b"], P[tb + ".ff.net.2.w"], P[tb + ".ff.net.2.b"], raw, t3, R=t2,
                       pre=pre, post=post)
        elif fold:
            gated = self.scratch("b", M, 4 * C)
            ops.gemm(t2, P[tb + ".ff1.w"], gated, bias=P[tb + ".ff1.c2"], act=L.ACT_GEGLU, C2=raw, lnfold=(rs, fold, P[tb + ".ff1.c1"], ls3, 1e-5))
            t3 = self.scratch("a", M, C)
            ops.gemm(gated, P[tb + ".ff.net.2.w"], t3, bias=P[tb + ".ff.net.2.b"], R=t2)
        elif fuse_ff:
            t3 = self.scratch("a2" if fuse_ln else "b", M, C)   # (not scratch "a": l3 is being read)
            ops.ff_fwd(l3, P[tb + ".ff1.w"], P[tb + ".ff1.b"], P[tb + ".ff.net.2.w"], P[tb + ".ff.net.2.b"], raw, t3, R=t2)
        else:
            gated = self.scratch("b", M, 4 * C)
            ops.gemm(l3, P[tb + ".ff1.w"], gated, bias=P[tb + ".ff1.b"], act=L.ACT_GEGLU, C2=raw)
            t3 = self.scratch("a", M, C)
            ops.gemm(gated, P[tb + ".ff.net.2.w"], t3, bias=P[tb + ".ff.net.2.b"], R=t2)
        if not (chain & 2):
            ops.gemm(t3, P[prefix + ".proj_out.w"], out, bias=P[prefix + ".proj_out.b"], R=x)
        stop_after_cross = (tb + ".attn2") == self.first_xattn

        def bwd(dout, dx):
            dt3 = self.scratch("g1", M, C)
            ops.gemm(dout, P[prefix + ".proj_out.wd"], dt3)
            dt2 = self.scratch("g3", M, C)
            if fuse_ff and FUSE_FF_LN:   # ... and the LayerNorm backward of norm3 in its epilogue (round 5): dt2 = LN'(dl3) + dt3 directly
                ops.ff_bwd(dt3, P[tb + ".ff.net.2.wd"], P[tb + ".ff1.wd"], raw, dt2, R=dt3, ln=(t2, ls3, ln_g("norm3")))
            elif fuse_ff:   # ff.net.2 dgrad, GEGLU backward and ff.net.0 dgrad in one launch; the LayerNorm backward behind it
                dl3 = self.scratch("g2", M, C)
                ops.ff_bwd(dt3, P[tb + ".ff.net.2.wd"], P[tb + ".ff1.wd"], raw, dl3)
                ops.layernorm_bwd(dl3, t2, ln_g("norm3"), ls3, dt2, add=dt3)
            else:
                dproj = self.scratch("gc", M, 8 * C)
                if FUSE_GEGLU_BWD or self.dtype == torch.float32:   # ff.net.2 dgrad with the GEGLU backward fused into its epilogue
                    ops.gemm(dt3, P[tb + ".ff.net.2.wd"], dproj, act=L.ACT_GEGLU_GRAD, C2=raw)
                else:
                    dgated = self.scratch("gb", M, 4 * C)
                    ops.gemm(dt3, P[tb + ".ff.net.2.wd"], dgated)
                    ops.geglu_bwd(dgated, raw, dproj)
                if fuse_ln:
                    ops.gemm(dproj, P[tb + ".ff1.wd"], dt2, R=dt3, ln_bwd=(P[tb + ".norm3.g"], ls3, t2))
                else:
                    dl3 = self.scratch("g2", M, C)
                    ops.gemm(dproj, P[tb + ".ff1.wd"], dl3)
                    ops.layernorm_bwd(dl3, t2, ln_g("norm3"), ls3, dt2, add=dt3)
            do2 = self.scratch("g1", M, C)
            ops.gemm(dt2, P[tb + ".attn2.to_out.0.wd"], do2)
            dq2 = self.scratch("g2", M, C)
            delta = self.scratch("delta", B * heads, HW, torch.float32)
            dk2, dv2 = self.dkv_all[:, ko:ko + C], self.dkv_all[:, ko + C:ko + 2 * C]
            xws = self.scratch("xattn_ws", 16 * 2 * B * T, C, torch.float32) if self.dtype != torch.float32 else None
            ops.attention_bwd(q2, k2, v2, o2, lse2, do2, delta, dq2, dk2, dv2, B, heads, HW, T, hd, ws=xws)
            if stop_after_cross:
                return
            dt1 = self.scratch("g4", M, C)
            if fuse_ln:
                ops.gemm(dq2, P[tb + ".attn2.to_q.wd"], dt1, R=dt2, ln_bwd=(P[tb + ".norm2.g"], ls2, t1))
            else:
                dl2 = self.scratch("g1", M, C)
                ops.gemm(dq2, P[tb + ".attn2.to_q.wd"], dl2)
                ops.layernorm_bwd(dl2, t1, ln_g("norm2"), ls2, dt1, add=dt2)
            do1 = self.scratch("g1", M, C)
            ops.gemm(dt1, P[tb + ".attn1.to_out.0.wd"], do1)
            dqkv = self.scratch("gq", M, 3 * C)
            # ws: 2 x [B, heads, HW] floats -- lets the hd = 40 / 64x64-map layers take the LDS-DMA staged dK/dV kernel (the dQ kernel
            # publishes -lse log2 e and -delta there for it); other shapes ignore it
            sws = self.scratch("sattn_ws", 2 * B * heads, HW, torch.float32) if hd in (40, 64, 80) and HW % 128 == 0 and self.dtype != torch.float32 else None
            ops.attention_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o1, lse1, do1, delta, dqkv[:, :C], dqkv[:, C:2 * C],
                              dqkv[:, 2 * C:], B, heads, HW, HW, hd, ws=sws)
            dt0 = self.scratch("g3", M, C)
            if fuse_ln:
                ops.gemm(dqkv, P[tb + ".attn1.qkv.wd"], dt0, R=dt1, ln_bwd=(P[tb + ".norm1.g"], ls1, t0))
            else:
                dl1 = self.scratch("g2", M, C)
                ops.gemm(dqkv, P[tb + ".attn1.qkv.wd"], dl1)
                ops.layernorm_bwd(dl1, t0, ln_g("norm1"), ls1, dt0, add=dt1)
            dn0 = self.scratch("g1", M, C)
            ops.gemm(dt0, P[prefix + ".proj_in.wd"], dn0)
            self._gn_bwd(dn0, x, prefix + ".norm", st0, dx, HW, False, add=dout)
        return bwd, stop_after_cross

    # ------------------------------------------------------------------ forward
    def _ensure_kv(self):
        if self._kv_pending is None:
            return
        (ehs16, ehs_ready), self._kv_pending = self._kv_pending, None
        if ehs_ready is not None:
            ehs_ready()   # e.g. torch.cuda.current_stream().wait_stream(text-encoder stream)
        B, P = self.B, self.P
        if getattr(self, "kv_r", 0) and self.dtype != torch.float32:   # bf16 mode: the same two products on the 16-bit operand copies
            self.kv_t = self.buf("kv_t", B * self.T, self.kv_n2r_pad)
            ops.gemm(ehs16, self.kv_A16, self.kv_t)
            ops.gemm(ehs16, P["kv_all.w"], self.kv_all, A2=self.kv_t, W2=self.kv_w2_16)
        elif getattr(self, "kv_r", 0):  # adapters: kv = ehs W^T + (ehs A_all^T) W2^T, W2 = block-structured scaling * B (pack_kv_lora)
            n2r = self.kv_w2.shape[1]
            self.kv_t = self.buf("kv_t", B * self.T, n2r)
            ops.gemm(ehs16, self.kv_lora_A.view(n2r, -1), self.kv_t)
            ops.gemm(ehs16, P["kv_all.w"], self.kv_all, A2=self.kv_t, W2=self.kv_w2)
        else:
            ops.gemm(ehs16, P["kv_all.w"], self.kv_all)

    def forward(self, sample, timesteps, ehs16, ehs_ready=None):
        """ehs_ready: optional callable run ONCE per forward, right before the first kernel that reads `ehs16` (the hoisted K/V projection, in front
        of the first cross-attention; at the end of the forward at the latest): a stream join when the caller produces `ehs16` on another stream
        while the layers in front of it already run, or the producer itself (the trainer issues the text-encoder forward from it)."""
        geo, B, P = self.geo, self.B, self.P
        ch = geo.block_out_channels
        nl = len(ch)
        hw = [(self.H >> l, self.W >> l) for l in range(nl)]
        Ms = [B * h * w for h, w in hw]
        L_ = geo.layers_per_block
        self.tape = []
        # ---- time embedding -> per-resnet row bias (one GEMM for all 22 time_emb_proj)
        te = self.buf("temb.sincos", B, ch[0])
        ops.timestep_embed(timesteps, te)
        te1 = self.buf("temb.l1", B, ch[0] * 4)
        ops.gemm(te, P["time_embedding.linear_1.w"], te1, bias=P["time_embedding.linear_1.b"], act=L.ACT_SILU)
        te2 = self.buf("temb.l2", B, ch[0] * 4)
        ops.gemm(te1, P["time_embedding.linear_2.w"], te2, bias=P["time_embedding.linear_2.b"], act=L.ACT_SILU)  # silu(temb)
        self.rowbias = self.buf("temb.rowbias", B, self.temb_total, torch.float32)
        ops.gemm(te2, P["temb_all.w"], self.rowbias, bias=P["temb_all.b"])
        # ---- hoisted cross-attention K/V projections of the text states: issued by `_ensure_kv` in front of the first cross-attention, the
        # first consumer of `ehs16` -- so that a caller whose text encoder still runs on another stream (`ehs_ready`) joins as late as possible
        self.kv_all = self.buf("kv_all", B * self.T, self.kv_total)
        self.dkv_all = self.buf("dkv_all", B * self.T, self.kv_total)
        self._kv_pending = (ehs16, ehs_ready)
        # ---- concat buffers of the up path (hidden part first, skip part second)
        skip_level = [0]
        for i in range(nl):
            skip_level += [i] * L_
            if i < nl - 1:
                skip_level.append(i + 1)
        nskips = len(self.skip_chs)
        cat_views = [None] * nskips          # skip k -> (cat buffer, h_cols)
        cats = {}
        sk = list(range(nskips))
        prev = ch[-1]
        for i, c in enumerate(reversed(ch)):
            for j in range(L_ + 1):
                k = sk.pop()
                hc = prev if j == 0 else c
                lvl = skip_level[k]
                cb = self.buf(f"cat.{i}.{j}", Ms[lvl], hc + self.skip_chs[k])
                cats[(i, j)] = (cb, hc)
                cat_views[k] = cb[:, hc:]
            prev = c
        # ---- down path
        x = cat_views[0]
        ops.conv4_to_nhwc(sample, P["conv_in.wp"], P["conv_in.b"], x, B, self.H, self.W, ch[0], sign=1)
        k = 1
        down_records = []   # (kind, bwd_fn / info, input_view, output_view, skip index of the output or None)
        # xpk: split-K slices of the convolution that produces x, when x itself is still unwritten -- the GroupNorm that opens the next block adds
        # them, applies the convolution's epilogue and writes x (`defer_out` / `x_partials`: only where that GroupNorm is the very next launch)
        xpk = None
        for i, c in enumerate(ch):
            for j in range(L_):
                has_attn = geo.cross_attn_levels[i]
                dst = cat_views[k]
                if has_attn:
                    mid = self.buf(f"down.{i}.{j}.res", Ms[i], c)
                    rb, pk = self._resnet(f"down_blocks.{i}.resnets.{j}", x, mid, hw[i], x_partials=xpk, defer_out=True)
                    tbwd, stop = self._transformer(f"down_blocks.{i}.attentions.{j}", mid, dst, hw[i], i, x_partials=pk)
                    xpk = None
                    down_records.append(("res", rb, x, mid, None))
                    down_records.append(("attn", (tbwd, stop), mid, dst, k))
                else:
                    # followed by the next resnet of the level, or (last level) by mid_block.resnets.0: both open with a GroupNorm over dst alone
                    rb, xpk = self._resnet(f"down_blocks.{i}.resnets.{j}", x, dst, hw[i], x_partials=xpk,
                                           defer_out=(j + 1 < L_ or i == nl - 1))
                    down_records.append(("res", rb, x, dst, k))
                x = dst
                k += 1
            if i < nl - 1:
                dst = cat_views[k]
                name = f"down_blocks.{i}.downsamplers.0.conv"
                xpk = self._pk(self._conv(x, name, dst, B, hw[i][0], hw[i][1], hw[i + 1][0], hw[i + 1][1], stride=2, bias=P[name + ".b"],
                                          defer=self._gn_splitk(hw[i + 1][0] * hw[i + 1][1], c)))
                down_records.append(("down", (name, i), x, dst, k))
                x = dst
                k += 1
        # ---- mid
        lvl = nl - 1
        m1 = self.buf("mid.r0", Ms[lvl], ch[-1])
        mb0, pk = self._resnet("mid_block.resnets.0", x, m1, hw[lvl], x_partials=xpk, defer_out=True)
        m2 = self.buf("mid.a0", Ms[lvl], ch[-1])
        mb1, _ = self._transformer("mid_block.attentions.0", m1, m2, hw[lvl], lvl, x_partials=pk)
        cb, hc = cats[(0, 0)]
        m3 = cb[:, :hc]
        mb2, _ = self._resnet("mid_block.resnets.1", m2, m3, hw[lvl])
        mid_records = [(mb0, x, m1), (mb1, m1, m2), (mb2, m2, m3)]
        # ---- up path
        up_records = []
        for i, c in enumerate(reversed(ch)):
            lvl = nl - 1 - i
            has_attn = geo.cross_attn_levels[lvl]
            for j in range(L_ + 1):
                cb, hc = cats[(i, j)]
                last = j == L_
                if not last:
                    nb, nhc = cats[(i, j + 1)]
                    dst = nb[:, :nhc]
                elif i < nl - 1:
                    dst = self.buf(f"up.{i}.out", Ms[lvl], c)
                else:
                    dst = self.buf("up.final", Ms[lvl], c)
                if has_attn:
                    mid = self.buf(f"up.{i}.{j}.res", Ms[lvl], c)
                    rb, pk = self._resnet(f"up_blocks.{i}.resnets.{j}", cb, mid, hw[lvl], defer_out=True)
                    tbwd, _ = self._transformer(f"up_blocks.{i}.attentions.{j}", mid, dst, hw[lvl], lvl, x_partials=pk)
                    up_records.append(("res", rb, cb, mid, (i, j)))
                    up_records.append(("attn", tbwd, mid, dst, None))
                else:
                    rb, _ = self._resnet(f"up_blocks.{i}.resnets.{j}", cb, dst, hw[lvl])
                    up_records.append(("res", rb, cb, dst, (i, j)))
                x = dst
            if i < nl - 1:
                nb, nhc = cats[(i + 1, 0)]
                dst = nb[:, :nhc]
                name = f"up_blocks.{i}.upsamplers.0.conv"
                if SUBPIXEL and (name + ".wsub") in P:
                    # four 2x2-tap convolutions on the COARSE map, each writing one parity class of the fine map (csrc/gemm8.hip, SUB = 1)
                    ops.gemm(x, P[name + ".wsub"], dst, bias=P[name + ".b"],
                             conv=dict(B=B, Hin=hw[lvl][0], Win=hw[lvl][1], Cin=x.shape[1], Hout=hw[lvl - 1][0], Wout=hw[lvl - 1][1], stride=1,
                                       sign=1, upsample=2, transposed=0))
                elif MATERIALIZE_UPSAMPLE and hw[lvl - 1][1] % 16 == 0:
                    # nearest x2 written out (12 us for the largest map), so that the halo-resident wide-tile kernel takes the convolution
                    # instead of the 4-wave gather with the upsampling folded in (311 -> ~205 us at 64x64 x 640 channels)
                    xu = self.scratch("up2x", Ms[lvl - 1], x.shape[1])
                    ops.upsample2x(x, xu, B, hw[lvl][0], hw[lvl][1], x.shape[1])
                    self._conv(xu, name, dst, B, hw[lvl - 1][0], hw[lvl - 1][1], hw[lvl - 1][0], hw[lvl - 1][1], bias=P[name + ".b"])
                else:
                    self._conv(x, name, dst, B, hw[lvl][0], hw[lvl][1], hw[lvl - 1][0], hw[lvl - 1][1], upsample=1, bias=P[name + ".b"])
                up_records.append(("up", (name, lvl), x, dst, None))
                x = dst
        # ---- head
        M0 = Ms[0]
        sto = self.buf("out.st", B * geo.norm_num_groups, 2, torch.float32)
        a = self.scratch("a", M0, ch[0])
        self._gn_fwd(x, "conv_norm_out", a, sto, hw[0][0] * hw[0][1], True)
        pred = self.buf("pred", B * 4, self.H * self.W).view(B, 4, self.H, self.W)
        ops.conv_to4(a, P["conv_out.wp"], P["conv_out.b"], pred, B, self.H, self.W, ch[0])
        self._saved = dict(final=x, sto=sto, cats=cats, cat_views=cat_views, down=down_records, mid=mid_records, up=up_records,
                           hw=hw, Ms=Ms, ehs16=ehs16)
        # `ehs_ready` may carry compute (the trainer issues the whole text-encoder forward from it, TextBoostStep.te_fwd_late): a geometry
        # without any cross-attention layer never reached `_ensure_kv` above -- run it here so that the callback is guaranteed to have happened
        self._ensure_kv()
        return pred

    # ------------------------------------------------------------------ backward (dgrad only, down to d_ehs)
    def backward(self, dpred, d_ehs_out=None):
        S, geo, B, P = self._saved, self.geo, self.B, self.P
        ch = geo.block_out_channels
        nl = len(ch)
        hw, Ms = S["hw"], S["Ms"]
        L_ = geo.layers_per_block
        # head
        da = self.scratch("g1", Ms[0], ch[0])
        ops.conv4_to_nhwc(dpred, P["conv_out.wdp"], None, da, B, self.H, self.W, ch[0], sign=-1)
        g = self.buf("grad.final", Ms[0], ch[0])
        self._gn_bwd(da, S["final"], "conv_norm_out", S["sto"], g, hw[0][0] * hw[0][1], True)
        # up path in reverse; d(cat) buffers are kept for the skip gradients
        dcat = {}
        for kind, fn, xin, xout, tag in reversed(S["up"]):
            M, cin = xin.shape
            if kind == "up":
                name, lvl = fn
                gx = self.buf(f"grad.up.{name}", M, cin)
                if SUBPIXEL and (name + ".wdsub") in self.P:
                    # dgrad of the sub-pixel convolutions: the fine gradient read as four strided views, 16 taps per coarse pixel (SUB = 2)
                    ops.gemm(g, self.P[name + ".wdsub"], gx,
                             conv=dict(B=B, Hin=hw[lvl - 1][0], Win=hw[lvl - 1][1], Cin=g.shape[1], Hout=hw[lvl][0], Wout=hw[lvl][1], stride=1,
                                       sign=1, upsample=3, transposed=0))
                else:
                    du = self.scratch("gu", Ms[lvl - 1], xin.shape[1])
                    self._conv(g, name, du, B, hw[lvl - 1][0], hw[lvl - 1][1], hw[lvl - 1][0], hw[lvl - 1][1], dgrad=True)
                    ops.pool2x2_sum(du, gx, B, hw[lvl][0], hw[lvl][1], cin)
                g = gx
            elif kind == "attn":
                gx = self.buf(f"grad.attn.{M}x{cin}.u", M, cin)
                fn(g, gx)
                g = gx
            else:
                i, j = tag
                gx = self.buf(f"grad.cat.{i}.{j}", M, cin)
                fn(g, gx)
                dcat[(i, j)] = gx
                hc = S["cats"][(i, j)][1]
                g = gx[:, :hc]
        # mid
        for mi, (fn, xin, xout) in reversed(list(enumerate(S["mid"]))):
            gx = self.buf(f"grad.mid.{mi}", xin.shape[0], xin.shape[1])
            fn(g, gx)
            g = gx
        # skip gradient views, indexed by skip number
        sk = list(range(len(self.skip_chs)))
        skip_grad = {}
        for i in range(nl):
            for j in range(L_ + 1):
                k = sk.pop()
                hc = S["cats"][(i, j)][1]
                skip_grad[k] = dcat[(i, j)][:, hc:]
        # down path in reverse: gradient of each skip tensor = grad from its down-path consumer + skip slice
        # g currently = gradient w.r.t. the last skip tensor coming from the mid block
        recs = list(reversed(S["down"]))
        premerged = False  # the previous record's dgrad convolution already added this record's skip gradient (its epilogue residual)
        for ri, (kind, fn, xin, xout, s_idx) in enumerate(recs):
            if s_idx is not None and not premerged:
                merged = self.buf(f"grad.skipmerge.{s_idx}", xout.shape[0], xout.shape[1])
                ops.add_f16(g, skip_grad[s_idx], merged)
                g = merged
            premerged = False
            if kind == "attn":
                tbwd, stop = fn
                if stop:
                    tbwd(g, None)   # first cross-attention: only dK/dV are produced; nothing upstream needs a gradient
                    break
                gx = self.buf(f"grad.attn.{xin.shape[0]}x{xin.shape[1]}.d", xin.shape[0], xin.shape[1])
                tbwd(g, gx)
                g = gx
            elif kind == "res":
                gx = self.buf(f"grad.res.{xin.shape[0]}x{xin.shape[1]}.d", xin.shape[0], xin.shape[1])
                fn(g, gx)
                g = gx
            else:
                name, lvl = fn
                gx = self.buf(f"grad.down.{lvl}", xin.shape[0], xin.shape[1])
                nxt = recs[ri + 1][4] if ri + 1 < len(recs) else None   # the layer in front of the downsampler wrote a skip tensor: its
                R = skip_grad[nxt] if nxt is not None else None         # gradient rides in as the dgrad GEMM's residual (no add launch)
                if (name + ".wdsub2") in P:
                    ops.gemm(g, P[name + ".wdsub2"], gx, R=R,
                             conv=dict(B=B, Hin=hw[lvl + 1][0], Win=hw[lvl + 1][1], Cin=g.shape[1], Hout=hw[lvl][0], Wout=hw[lvl][1], stride=1, sign=1,
                                       upsample=2, transposed=0))
                else:
                    self._conv(g, name, gx, B, hw[lvl + 1][0], hw[lvl + 1][1], hw[lvl][0], hw[lvl][1], dgrad=True, transposed=1, R=R)
                premerged = R is not None
                g = gx
        # hoisted K/V dgrad -> d encoder_hidden_states (fp32)
        if d_ehs_out is None:
            d_ehs_out = self.buf("d_ehs", B * self.T, geo.cross_attention_dim, torch.float32)
        ops.gemm(self.dkv_all, P["kv_all.wd"], d_ehs_out)
        if getattr(self, "kv_r", 0):
            # adapter gradients off the hoisted operand (ACCUMULATED into kv_grad_A / kv_grad_B: the trainer zeroes the flat gradient buffer):
            #   dt = dkv W2                       dB[rows of (layer, proj)] += scaling * dkv[:, rows]^T t[:, cols]
            #   dA_all += dt^T ehs                d_ehs += dt A_all
            M, r, n2r, Dc = B * self.T, self.kv_r, self.kv_w2.shape[1], geo.cross_attention_dim
            ehs, dkv, kv_t = S["ehs16"], self.dkv_all, self.kv_t
            if self.dtype != torch.float32:   # bf16 mode: the fp32 path's (exact-fp32 MFMA) products on fp32 copies of the 16-bit operands
                dkv = self.buf("kv_dkv32", M, self.kv_total, torch.float32)
                ops.convert(self.dkv_all, dkv)
                kv_t = self.buf("kv_t32", M, n2r, torch.float32)
                ops.convert(self.kv_t[:, :n2r], kv_t)
                e32 = self.buf("kv_ehs32", M, Dc, torch.float32)
                ops.convert(ehs, e32)
                ehs = e32
            dt = self.buf("kv_dt", M, n2r, torch.float32)
            ops.gemm_f32_t(dkv, self.kv_w2, dt, M, n2r, self.kv_total, w_trans=True)
            for l, (p, C) in enumerate(self.xattn):
                ko = self.kv_off[p]
                for proj in range(2):
                    rows = slice(ko + proj * C, ko + (proj + 1) * C)
                    cols = slice(l * 2 * r + proj * r, l * 2 * r + (proj + 1) * r)
                    ops.gemm_f32_t(dkv[:, rows], kv_t[:, cols], self.kv_grad_B[rows], C, r, M, a_trans=True, w_trans=True,
                                   R=self.kv_grad_B[rows], alpha=self.kv_scaling)
            gA = self.kv_grad_A.view(n2r, Dc)
            ops.gemm_f32_t(dt, ehs, gA, n2r, Dc, M, a_trans=True, w_trans=True, R=gA)
            ops.gemm_f32_t(dt, self.kv_lora_A.view(n2r, Dc), d_ehs_out, M, Dc, n2r, w_trans=True, R=d_ehs_out)
        return d_ehs_out
