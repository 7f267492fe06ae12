"""HIP executor of the TextBoost text encoder: CLIP text transformer + rank-r LoRA on q/k/v + TextBoost pins.

Mirrors /root/reference/textboost/text_encoder.py:17-87 (`TextBoostModel`), the LoRA injection of
train_textboost.py:700-722 (peft `LoraConfig(r, lora_alpha=r, init_lora_weights="gaussian",
target_modules=["q_proj","k_proj","v_proj"])`) and the token-table growth of textboost/utils.py:117-166.

Three numeric modes, matching how the reference runs the two encoders:
  * "autocast" -- the trainable encoder under accelerate autocast (--mixed_precision fp16): fp32 master params, fp32 residual stream and
    LayerNorm/softmax statistics, fp16 Linear/attention operands (`train_textboost.py:919-926`);
  * "half"     -- the KPL teacher `original_text_encoder.to(fp16)` (:650, :939): plain fp16 module, forward only;
  * "fp32"     -- no mixed precision (the default, :298-308; weight_dtype stays float32, :930-939): every parameter, activation
    and gradient fp32, for the trainable encoder and the teacher alike (csrc/f32_path.hip).

All arithmetic is in libtextboost_hip.so; this file owns parameters, buffers and the layer schedule.
Trainable state lives in three flat fp32 buffers so the optimizer / all-reduce see one tensor each:
  lora_A [L, 3r, D], lora_B [L, 3D, r]  and the token table [V + k, D] (rows >= first_added train).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops

EOS_ID = 49407  # textboost/text_encoder.py:71
CHAIN_LORA_BWD = os.environ.get("TB_CHAIN_LORA_BWD", "1") != "0"   # A/B knob: 0 = one dt/dB + one dA launch per layer


@dataclass
class CLIPGeometry:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_layers: int = 12
    num_heads: int = 12
    max_pos: int = 77
    act: str = "quick_gelu"
    eps: float = 1e-5


HF_LAYER = {"ln1": "layer_norm1", "ln2": "layer_norm2", "q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj",
            "out": "self_attn.out_proj", "fc1": "mlp.fc1", "fc2": "mlp.fc2"}


class HipTextEncoder:
    def __init__(self, geo: CLIPGeometry, state_dict: Dict[str, torch.Tensor], batch: int, mode: str = "autocast", lora_rank: int = 0,
                 lora_alpha: Optional[float] = None, n_slots: int = 1, device="cuda", seed: Optional[int] = None):
        """state_dict uses transformers CLIPTextModel keys (`text_model.embeddings.token_embedding.weight`, ...)."""
        assert mode in ("autocast", "half", "fp32")
        assert geo.act in ("quick_gelu", "gelu")  # SD1.x CLIP-L / SD2.x OpenCLIP-H
        self.act_fwd, self.act_bwd = (L.ACT_QUICK_GELU, L.ACT_QUICK_GELU_GRAD) if geo.act == "quick_gelu" else (L.ACT_GELU, L.ACT_GELU_GRAD)
        self.geo, self.B, self.T, self.mode, self.dev = geo, batch, geo.max_pos, mode, device
        self.r = lora_rank
        self.scaling = (lora_alpha if lora_alpha is not None else lora_rank) / lora_rank if lora_rank else 0.0
        self.res_dtype = torch.float32 if mode in ("autocast", "fp32") else L.half_dtype()
        self.op_dtype = torch.float32 if mode == "fp32" else L.half_dtype()   # Linear / attention operands and activations
        self.n_slots = n_slots
        self._bufs: Dict[str, torch.Tensor] = {}
        D, Lr = geo.hidden_size, geo.num_layers
        sd = state_dict
        pre = "text_model."
        odt = self.op_dtype
        f16 = lambda t: t.detach().to(odt).to(device).contiguous()
        f32r = lambda t: t.detach().to(odt).to(torch.float32).to(device).contiguous()  # rounded like the operands (not at all in fp32 mode), fp32 storage
        tbl_dt = torch.float32 if mode in ("autocast", "fp32") else L.half_dtype()
        self.token_table = sd[pre + "embeddings.token_embedding.weight"].detach().to(tbl_dt).to(device).contiguous()
        self.pos_table = sd[pre + "embeddings.position_embedding.weight"].detach().to(tbl_dt).to(device).contiguous()
        # LayerNorm affine: fp32 params under autocast (LN runs in fp32); fp16-rounded for the half module
        lnp = (lambda t: t.detach().float().to(device).contiguous()) if mode in ("autocast", "fp32") else f32r
        self.Wl: List[Dict[str, torch.Tensor]] = []
        for i in range(Lr):
            lp = f"{pre}encoder.layers.{i}."
            W = {}
            wq = torch.cat([sd[lp + HF_LAYER[x] + ".weight"] for x in "qkv"], dim=0)
            W["qkv.w"], W["qkv.wd"] = f16(wq), f16(wq).t().contiguous()
            W["qkv.b"] = f32r(torch.cat([sd[lp + HF_LAYER[x] + ".bias"] for x in "qkv"], dim=0))
            for n in ("out", "fc1", "fc2"):
                w = sd[lp + HF_LAYER[n] + ".weight"]
                W[n + ".w"], W[n + ".wd"] = f16(w), f16(w).t().contiguous()
                W[n + ".b"] = f32r(sd[lp + HF_LAYER[n] + ".bias"])
            for n in ("ln1", "ln2"):
                W[n + ".g"], W[n + ".b"] = lnp(sd[lp + HF_LAYER[n] + ".weight"]), lnp(sd[lp + HF_LAYER[n] + ".bias"])
            self.Wl.append(W)
        self.lnf_g, self.lnf_b = lnp(sd[pre + "final_layer_norm.weight"]), lnp(sd[pre + "final_layer_norm.bias"])
        if mode == "fp32":
            for slot in (0, 1, 2):  # (an encoder may run on the trainer's side streams: one score workspace per stream slot, sized before any capture)
                with ops.workspace_slot(slot):
                    ops.reserve_attention_f32(device, 3 * batch, geo.num_heads, self.T, self.T)   # student + prior + teacher rows in one pass
        self.null_embedding = torch.zeros(self.T, D, device=device, dtype=torch.float32)
        self.use_fixed_special_embedding = False
        self.first_added = self.token_table.shape[0]
        self.n_added = 0
        if self.r:
            g = torch.Generator(device="cpu")
            if seed is not None:
                g.manual_seed(seed)
            # peft init_lora_weights="gaussian": A ~ N(0, (1/r)^2), B = 0
            self.lora_A = (torch.randn(Lr, 3 * self.r, D, generator=g) / self.r).to(device)
            self.lora_B = torch.zeros(Lr, 3 * D, self.r, device=device)
            self.grad_A = torch.zeros_like(self.lora_A)
            self.grad_B = torch.zeros_like(self.lora_B)
            self.w2_fwd = torch.zeros(Lr, 3 * D, 64, device=device, dtype=odt)
            self.w2_dgrad = torch.zeros(Lr, D, 64, device=device, dtype=odt)

    # ------------------------------------------------------------------ reference-facing surface
    def set_null_embedding(self, null):  # text_encoder.py:28-32
        self.null_embedding = null.detach().to(torch.float32).to(self.dev).contiguous()
        self.use_fixed_special_embedding = True

    def get_input_embeddings_weight(self):
        return self.token_table

    def add_tokens(self, initializer_ids):
        """textboost/utils.py:117-166 row bookkeeping: grow the table, copy initializer rows; returns new ids."""
        V, D = self.token_table.shape
        k = len(initializer_ids)
        new = torch.empty(V + k, D, device=self.dev, dtype=self.token_table.dtype)
        new[:V] = self.token_table
        for j, init in enumerate(initializer_ids):
            new[V + j] = self.token_table[init]
        self.token_table = new
        if self.n_added == 0:
            self.first_added = V
        self.n_added += k
        self.grad_added = torch.zeros(self.n_added, D, device=self.dev)
        return list(range(V, V + k))

    def __call__(self, input_ids, attention_mask=None, return_dict=False, slot=0):
        assert attention_mask is None, "text_encoder_use_attention_mask is off in the reference defaults (utils.py:14-17)"
        return (self.forward(input_ids, slot=slot), None)

    # ------------------------------------------------------------------ buffers
    def buf(self, name, rows, cols, dtype):
        """named, zero-initialised activation buffer.  A buffer that already exists with MORE rows is returned as its leading row slice:
        the forward with a frozen extra batch (`extra_ids`) sizes everything for student + extra rows, the backward walks the student rows."""
        t = self._bufs.get(name)
        if t is None or t.shape[1] != cols or t.dtype != dtype or t.shape[0] < rows:
            t = torch.zeros(rows, cols, device=self.dev, dtype=dtype)
            self._bufs[name] = t
        return t if t.shape[0] == rows else t[:rows]

    def pack_lora(self):
        """refresh the fp16 K-extension operands from the fp32 LoRA masters (once per optimizer step)."""
        if not self.r:
            return
        D = self.geo.hidden_size
        assert self.lora_A.is_contiguous() and self.lora_B.is_contiguous()
        ops.lora_pack(self.lora_A, self.lora_B, self.w2_fwd, self.w2_dgrad, D, D, self.r, 3, self.scaling, layers=self.geo.num_layers)

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, slot=0, pins=True, extra_ids=None, extra_table=None):
        """extra_ids [Be, T] (+ extra_table, the ORIGINAL fp32 token table): a frozen batch that rides along in the same launches -- its rows
        get no LoRA term (their K-extension operand rows stay zero) and are embedded from `extra_table`; used for the KPL teacher
        (train_textboost.py:1096-1100: the un-adapted encoder on the prior prompts), whose separate M = 616 pass was pure launch latency.
        Returns [(B + Be) * T, D]: student rows first.  backward() only ever walks the student rows."""
        geo, B, T = self.geo, input_ids.shape[0], self.T
        D, I, H = geo.hidden_size, geo.intermediate_size, geo.num_heads
        hd = D // H
        Bs, Ms = B, B * T                # student sequences / rows (LoRA, trainable table)
        if extra_ids is not None:
            B = B + extra_ids.shape[0]
        M = B * T
        rdt, f16, f32 = self.res_dtype, self.op_dtype, torch.float32   # (`f16` = the operand dtype: fp32 in the no-AMP mode)
        full32 = self.mode == "fp32"
        ids = input_ids.reshape(-1).contiguous()
        s = f"s{slot}."
        self._bufs[s + "ids"] = ids
        h = self.buf(s + "h0", M, D, rdt)
        ops.embed_fwd(ids, self.token_table, self.pos_table, h[:Ms], T)
        if extra_ids is not None:
            eids = extra_ids.reshape(-1).contiguous()
            ops.embed_fwd(eids, extra_table, self.pos_table, h[Ms:], T)
        for i, W in enumerate(self.Wl):
            p = f"{s}l{i}."
            x1 = self.buf(p + "x1", M, D, f16)
            ls1 = self.buf(p + "ls1", M, 2, f32)
            qkv = self.buf(p + "qkv", M, 3 * D, f16)
            if self.r:
                t = self.buf(p + "t", M, 64, f16)  # columns >= 3r (and the frozen extra rows) stay zero (K-extension operand of the qkv GEMM)
                if full32:  # no fused (fp16) down projection: t[:Ms, :3r] = x1 @ A^T as an exact-fp32 GEMM
                    ops.layernorm_fwd(h, x1, W["ln1.g"], W["ln1.b"], ls1, geo.eps)
                    ops.gemm(x1[:Ms], self.lora_A[i], t[:Ms, :3 * self.r])
                else:
                    ops.layernorm_fwd(h, x1, W["ln1.g"], W["ln1.b"], ls1, geo.eps, lora_A=self.lora_A[i], t=t, lora_rows=Ms)
                ops.gemm(x1, W["qkv.w"], qkv, A2=t, W2=self.w2_fwd[i], bias=W["qkv.b"])
            else:
                ops.layernorm_fwd(h, x1, W["ln1.g"], W["ln1.b"], ls1, geo.eps)
                ops.gemm(x1, W["qkv.w"], qkv, bias=W["qkv.b"])
            o = self.buf(p + "o", M, D, f16)
            lse = self.buf(p + "lse", B * H, T, f32)
            ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, B, H, T, T, hd, causal=True)
            h2 = self.buf(p + "h2", M, D, rdt)
            ops.gemm(o, W["out.w"], h2, bias=W["out.b"], R=h)
            x2 = self.buf(s + "x2", M, D, f16)
            ls2 = self.buf(p + "ls2", M, 2, f32)
            ops.layernorm_fwd(h2, x2, W["ln2.g"], W["ln2.b"], ls2, geo.eps)
            a = self.buf(s + "a", M, I, f16)
            pre = self.buf(p + "pre", M, I, f16)
            ops.gemm(x2, W["fc1.w"], a, bias=W["fc1.b"], act=self.act_fwd, C2=pre)
            h3 = self.buf(p + "h3", M, D, rdt)
            ops.gemm(a, W["fc2.w"], h3, bias=W["fc2.b"], R=h2)
            h = h3
        out = self.buf(s + "out", M, D, rdt)
        lsf = self.buf(s + "lsf", M, 2, f32)
        ops.layernorm_fwd(h, out, self.lnf_g, self.lnf_b, lsf, geo.eps)
        if pins:
            ops.pin_fwd(out[:Ms], ids, self.null_embedding, Bs, T, self.use_fixed_special_embedding, EOS_ID)
            if M > Ms:
                ops.pin_fwd(out[Ms:], eids, self.null_embedding, B - Bs, T, self.use_fixed_special_embedding, EOS_ID)
        return out

    # ------------------------------------------------------------------ backward (autocast mode only)
    def zero_grad(self):
        if self.r:
            self.grad_A.zero_()
            self.grad_B.zero_()
        if self.n_added:
            self.grad_added.zero_()

    def backward(self, d_out, slot=0, pins=True, grads=None):
        """d_out: fp32 [B*T, D] gradient of the (scaled) loss w.r.t. forward(slot)'s output. Accumulates into
        grad_A / grad_B / grad_added -- or into `grads` = (grad_A, grad_B, grad_added) when given: two backward passes that run
        CONCURRENTLY on two streams (the trainer's instance-prompt and prior-prompt chains) must not add into the same buffers.
        All scratch is keyed by `slot` for the same reason."""
        assert self.mode in ("autocast", "fp32")
        grad_A, grad_B, grad_added = grads if grads is not None else (self.grad_A if self.r else None, self.grad_B if self.r else None,
                                                                      self.grad_added if self.n_added else None)
        full32 = self.mode == "fp32"
        geo, T = self.geo, self.T
        D, I, H = geo.hidden_size, geo.intermediate_size, geo.num_heads
        hd = D // H
        s = f"s{slot}."
        ids = self._bufs[s + "ids"]
        M = ids.numel()
        B = M // T
        f16, f32 = self.op_dtype, torch.float32
        gs = f"g{slot}."
        if pins:
            ops.pin_bwd(d_out, ids, B, T, self.use_fixed_special_embedding, EOS_ID)
        h_last = self._bufs[f"{s}l{geo.num_layers - 1}.h3"][:M]
        dh = self.buf(gs + "dh_a", M, D, f32)
        # fp16 copy of the running residual gradient, written by the LayerNorm backward (fp32 mode: the fp32 gradient itself feeds the GEMMs)
        dh16 = None if full32 else self.buf(gs + "dh16", M, D, f16)
        ops.layernorm_bwd(d_out, h_last, self.lnf_g, self._bufs[s + "lsf"], dh, dx16=dh16)
        dh_other = self.buf(gs + "dh_b", M, D, f32)
        pending = None
        for i in reversed(range(geo.num_layers)):
            W = self.Wl[i]
            p = f"{s}l{i}."
            # saved activations: the leading M rows (the forward may have carried a frozen extra batch behind them)
            h_in = (self._bufs[f"{s}l{i - 1}.h3"] if i > 0 else self._bufs[s + "h0"])[:M]
            h2, pre, qkv, o = (self._bufs[p + n][:M] for n in ("h2", "pre", "qkv", "o"))
            lse = self._bufs[p + "lse"][:B * H]
            x1 = self._bufs[p + "x1"][:M]
            dpre = self.buf(gs + "dpre", M, I, f16)
            ops.gemm(dh if full32 else dh16, W["fc2.wd"], dpre, act=self.act_bwd, C2=pre)
            dx2 = self.buf(gs + "dx", M, D, f16)
            ops.gemm(dpre, W["fc1.wd"], dx2)
            dh2 = dh_other
            ops.layernorm_bwd(dx2, h2, W["ln2.g"], self._bufs[p + "ls2"], dh2, add=dh, dx16=dh16)
            do = self.buf(gs + "do", M, D, f16)
            ops.gemm(dh2 if full32 else dh16, W["out.wd"], do)
            dqkv = self.buf(gs + "dqkv", M, 3 * D, f16)
            delta = self.buf(gs + "delta", B * H, T, f32)
            ops.attention_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, lse, do, delta, dqkv[:, :D], dqkv[:, D:2 * D],
                              dqkv[:, 2 * D:], B, H, T, T, hd, causal=True)
            dx1 = self.buf(gs + "dx", M, D, f16)
            if self.r:
                if full32 or not CHAIN_LORA_BWD:
                    dt = self.buf(gs + "dt", M, 64, f16)
                    ops.lora_bwd(dqkv, x1, self._bufs[p + "t"][:M], self.lora_B[i], dt, grad_A[i], grad_B[i], D, D, self.r, 3,
                                 self.scaling, w2_fwd=self.w2_fwd[i])
                else:
                    # chained: this layer's dA panels (dt^T x1, needed by nothing in the backward) ride in the NEXT layer's dt / dB launch,
                    # which leaves a third of the chip idle -- 11 launches fewer; two dt buffers alternate
                    dt = self.buf(gs + f"dt{i & 1}", M, 64, f16)
                    pending = ops.lora_bwd(dqkv, x1, self._bufs[p + "t"][:M], self.lora_B[i], dt, grad_A[i], grad_B[i], D, D, self.r, 3,
                                           self.scaling, pending=pending, defer_da=i > 0)
                ops.gemm(dqkv, W["qkv.wd"], dx1, A2=dt, W2=self.w2_dgrad[i])
            else:
                ops.gemm(dqkv, W["qkv.wd"], dx1)
            ops.layernorm_bwd(dx1, h_in, W["ln1.g"], self._bufs[p + "ls1"], dh, add=dh2, dx16=dh16 if (i > 0 and not full32) else None)
            # dh (buffer a) now holds the gradient w.r.t. this layer's input; dh2 (buffer b) is free again
        if self.n_added:
            ops.embed_bwd(dh, ids, grad_added, self.first_added)
        return dh
