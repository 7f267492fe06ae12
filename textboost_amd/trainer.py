"""One TextBoost optimizer step on MI355X -- the body of the reference's hot loop, train_textboost.py:1040-1149,
with every device op replaced by the HIP kernels behind libtextboost_hip.so and NO host synchronisation:

    noise / timesteps (torch generators, :1041-1048)  ->  add_noise (:1052)  ->  text encoder (:1054-1059)
    ->  UNet (:1063-1067)  ->  MSE (:1085-1090)  ->  KPL (:1096-1106)  ->  backward (:1108)  ->  grad row mask (:1109-1117)
    ->  [RCCL all-reduce of the flat trainable-gradient buffer (DDP, :919-926)]
    ->  unscale + clip (:1128-1133)  ->  AdamW x2 groups (:1134)  ->  added-row renorm (:1138-1149)

The whole step has static shapes and is captured into a HIP graph (`capture()`); `replay()` re-runs it.
Scalars (loss, loss scale, grad norm, found_inf, step count) stay on the device in `state` (fp32[16]).
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib as L
from . import ops
from .text_encoder import HipTextEncoder
from .unet import HipUNet


@dataclass
class StepHyper:
    lr: float = 5e-5              # --learning_rate
    emb_lr: float = 1e-3          # --emb_learning_rate
    beta1: float = 0.9
    beta2: float = 0.999
    wd: float = 1e-2              # --adam_weight_decay (:245)
    eps: float = 1e-8
    max_grad_norm: float = 1.0
    kpl_weight: float = 0.1       # :115
    kpl_type: str = "cos"         # :116 ("cos" | "mse")
    mixing: Optional[str] = None  # --mixing with --augment_ops object|style: zero odd / even rows of every lora_B.grad (:1119-1126)
    prediction_type: str = "epsilon"
    use_grad_scaler: bool = True  # --mixed_precision=fp16 (accelerate GradScaler)
    init_scale: float = 65536.0
    growth_interval: int = 2000
    num_train_timesteps: int = 1000
    grad_accum: int = 1           # --gradient_accumulation_steps (:1039 accelerator.accumulate; single process only, :573-577)


def alphas_cumprod(T=1000, beta_start=0.00085, beta_end=0.012, device="cuda"):
    """DDPMScheduler(beta_schedule="scaled_linear") of SD (diffusers), computed like diffusers in fp32."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).to(device)


def sum_gradients(flat_grad: torch.Tensor, world_size: int, force: bool = False):
    """The exchange step of DDP (:919-926): ONE all-reduce (SUM) of the flat trainable-gradient buffer.  DDP's division by the world size
    is not a pass over the buffer: `tb_scaler_update(grad_div=W)` folds it into the unscale / clip coefficients the optimizer kernels
    multiply the gradients with.  Row masking (:1114-1117) commutes with the mean, so reducing only the k added rows + the LoRA tensors is
    identical to the reference's dense reduction.  Backend-agnostic (RCCL on GPUs -- capturable in a HIP graph --, gloo in the CPU tests)."""
    if world_size > 1 or force:
        import torch.distributed as dist
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def average_gradients(flat_grad: torch.Tensor, world_size: int, force: bool = False):
    """sum over ranks, then 1/W: the gradient DDP hands to the optimizer (reference semantics; host-side helper for tests / tools -- the
    step itself uses `sum_gradients` + the folded division)."""
    sum_gradients(flat_grad, world_size, force)
    if world_size > 1 or force:
        flat_grad.mul_(1.0 / world_size)
    return flat_grad


def lr_lambda(name: str, num_warmup_steps: int, num_training_steps: int, lr_init: float = 1.0, num_cycles=None, power: float = 1.0,
              lr_end: float = 1e-7):
    """The multiplier lambda(step) of diffusers.optimization.get_scheduler(name, ...) (train_textboost.py:911-916; the same schedule
    family as transformers.optimization).  step = number of `lr_scheduler.step()` calls so far (0 for the first optimizer step)."""
    import math
    W, T = num_warmup_steps, num_training_steps

    def warm(step):
        return float(step) / float(max(1, W))

    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: warm(step) if step < W else 1.0
    if name == "linear":
        return lambda step: warm(step) if step < W else max(0.0, float(T - step) / float(max(1, T - W)))
    if name == "cosine":
        nc = 0.5 if num_cycles is None else num_cycles

        def f(step):
            if step < W:
                return warm(step)
            pr = float(step - W) / float(max(1, T - W))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(nc) * 2.0 * pr)))
        return f
    if name == "cosine_with_restarts":
        nc = 1 if num_cycles is None else num_cycles

        def f(step):
            if step < W:
                return warm(step)
            pr = float(step - W) / float(max(1, T - W))
            if pr >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(nc) * pr) % 1.0))))
        return f
    if name == "polynomial":
        def f(step):
            if step < W:
                return warm(step)
            if step > T:
                return lr_end / lr_init
            decay = (lr_init - lr_end) * (1 - (step - W) / (T - W)) ** power + lr_end
            return decay / lr_init
        return f
    raise ValueError(f"unknown lr_scheduler {name}")


def shard_indices(n_samples: int, batch: int, it: int, rank: int, world: int):
    """data-parallel sample assignment: global sample (it*W + rank)*B + b, wrapped over the dataset, so every rank's shard is
    non-empty even with ONE training image (the reference's Wrapper hangs for rank >= 1 there)."""
    return [((it * world + rank) * batch + b) % n_samples for b in range(batch)]


class TextBoostStep:
    def __init__(self, unet: HipUNet, text_encoder: HipTextEncoder, teacher: Optional[HipTextEncoder], hyper: StepHyper,
                 latent_shape, device="cuda", world_size: int = 1, generator: Optional[torch.Generator] = None):
        self.unet, self.te, self.teacher, self.hp, self.dev = unet, text_encoder, teacher, hyper, device
        self.world = world_size
        self.force_dist = False  # tests: exercise the collective + two-graph path with a 1-rank process group
        self.gen = generator
        B, C, H, W = latent_shape
        self.B = B
        te = text_encoder
        D = te.geo.hidden_size
        self.acp = alphas_cumprod(hyper.num_train_timesteps, device=device)
        # ---- flat trainable-gradient buffer: [grad_A | grad_B | grad_added]  (one all-reduce, one sumsq each)
        has_lora = te.r > 0  # --lora_rank 0: only the added token rows train (:700: no adapter is injected, optimizer group 1 is empty)
        nA, nB, nE = (te.lora_A.numel(), te.lora_B.numel(), te.n_added * D) if has_lora else (0, 0, te.n_added * D)
        # --unet_params_to_train crossattn_kv (:712-721, HipUNet.enable_kv_lora): the UNet's K/V adapters are the optimizer's THIRD group
        # (:838-841: default lr, same decay; the clip at :1128-1133 covers the text encoder only); their gradients ride in the same flat buffer
        nUA = unet.kv_lora_A.numel() if getattr(unet, "kv_r", 0) else 0
        nUB = unet.kv_lora_B.numel() if nUA else 0
        self.n_unet = nUA + nUB
        self.flat_grad = torch.zeros(nA + nB + nE + self.n_unet, device=device)
        if has_lora:
            self.flat_lora = torch.cat([te.lora_A.reshape(-1), te.lora_B.reshape(-1)])  # fp32 masters, flat
            te.lora_A = self.flat_lora[:nA].view_as(te.lora_A)
            te.lora_B = self.flat_lora[nA:].view_as(te.lora_B)
            te.grad_A = self.flat_grad[:nA].view_as(te.lora_A)
            te.grad_B = self.flat_grad[nA:nA + nB].view_as(te.lora_B)
        else:
            self.flat_lora = torch.zeros(0, device=device)
        te.grad_added = self.flat_grad[nA + nB:nA + nB + nE].view(te.n_added, D)
        self.n_lora = nA + nB
        self.n_emb = nE
        if self.n_unet:
            o = nA + nB + nE
            self.flat_unet = torch.cat([unet.kv_lora_A.reshape(-1), unet.kv_lora_B.reshape(-1)])
            unet.kv_lora_A = self.flat_unet[:nUA].view_as(unet.kv_lora_A)
            unet.kv_lora_B = self.flat_unet[nUA:].view_as(unet.kv_lora_B)
            unet.kv_grad_A = self.flat_grad[o:o + nUA].view_as(unet.kv_lora_A)
            unet.kv_grad_B = self.flat_grad[o + nUA:].view_as(unet.kv_lora_B)
            self.m_unet = torch.zeros(self.n_unet, device=device)
            self.v_unet = torch.zeros(self.n_unet, device=device)
        self.m_lora = torch.zeros(self.n_lora, device=device)
        self.v_lora = torch.zeros(self.n_lora, device=device)
        self.m_emb = torch.zeros(nE, device=device)
        self.v_emb = torch.zeros(nE, device=device)
        self.state = torch.zeros(L.ST_COUNT, device=device)
        self.state[L.ST_LOSS_SCALE] = hyper.init_scale if hyper.use_grad_scaler else 1.0
        # ---- mean_norm over ALL rows after token addition (:1017); one-time host read is outside the hot loop
        norms = torch.empty((te.token_table.shape[0] + 3) // 4 * 4, device=device)
        pad = norms.numel() - te.token_table.shape[0]
        if pad:
            tbl = torch.cat([te.token_table, torch.zeros(pad, D, device=device)])
            ops.row_norms(tbl, norms)
            self.mean_norm = float(norms[: te.token_table.shape[0]].mean().item())
        else:
            ops.row_norms(te.token_table, norms)
            self.mean_norm = float(norms.mean().item())
        # ---- static step I/O
        self.x0 = torch.zeros(B, C, H, W, device=device)
        self.noise = torch.zeros(B, C, H, W, device=device)
        self.timesteps = torch.zeros(B, dtype=torch.int64, device=device)
        # the instance prompts (:1054) and the KPL prior prompts (:1099) go through the trainable encoder as ONE batch of 2B
        # rows (rows are independent, so this is arithmetically identical to the reference's two calls)
        self.kpl = hyper.kpl_weight > 0 and teacher is not None
        nb = 2 * B if self.kpl else B
        self.ids_all = torch.zeros(nb, te.T, dtype=torch.int64, device=device)
        self.input_ids = self.ids_all[:B]
        self.prior_ids = self.ids_all[B:] if self.kpl else torch.zeros(B, te.T, dtype=torch.int64, device=device)
        self.d_all = torch.zeros(nb * te.T, D, device=device)
        # activations handed to the UNet: fp16 under --mixed_precision fp16 (`.to(unet.dtype)`, :1064-1066), fp32 in the no-AMP mode
        adt = unet.dtype
        assert (adt == torch.float32) == (text_encoder.mode == "fp32"), "fp32 (no-AMP) mode needs both the UNet and the text encoder in fp32"
        self.noisy = torch.empty(B, C, H, W, device=device, dtype=adt)
        self.velocity = torch.empty(B, C, H, W, device=device) if hyper.prediction_type == "v_prediction" else None
        self.dpred = torch.empty(B, C, H, W, device=device)
        self.ehs16 = torch.empty(B * te.T, D, device=device, dtype=adt)
        self.d_ehs = self.d_all[: B * te.T]
        self.d_prior = self.d_all[B * te.T:]
        self.kpl_partial = torch.empty(B * te.T, device=device)
        self.added_norms = torch.empty(max(te.n_added, 1), device=device)
        self.accum = max(1, int(hyper.grad_accum))
        assert self.accum == 1 or world_size == 1, "Gradient accumulation is not supported when training with multiple processes."  # :573-577
        self._micro = 0          # position inside the accumulation cycle
        self.graph = None
        self.graph_mode = "eager"
        self.external_noise = False
        self.side = torch.cuda.Stream(device=device) if self.kpl else None
        # The KPL teacher is the un-adapted encoder on the prior prompts (:1096-1100).  As its own M = 616 pass it was ~90 launches of pure launch
        # latency (0.87 ms); merged, its rows ride along in the student's launches with a zero LoRA operand and the original token table.
        # Arithmetic: the student's autocast path (fp32 residual stream) instead of the fp16 module's -- closer to fp32 arithmetic, not bit-equal
        # to the separate pass (TB_SEPARATE_TEACHER=1 keeps that pass).
        self.merge_teacher = self.kpl and text_encoder.r > 0 and os.environ.get("TB_SEPARATE_TEACHER", "0") != "1"
        self.teacher_table32 = teacher.token_table.float().contiguous() if self.merge_teacher else None
        # Split text-encoder schedule (round 3): the encoder's ~250 launches per step are latency-bound (a fraction of one chip round each), and
        # only the INSTANCE rows' forward feeds the UNet / only their backward waits for it.  The prior-prompt rows + the frozen teacher rows, the
        # KPL loss and the prior rows' whole backward run as a second branch of the step graph beside the UNet (branches of one HIP graph do run
        # concurrently on ROCm 7: scratch/graph_branches.py), and the instance rows' forward runs beside the UNet layers in front of the first
        # cross-attention.  Gradients of the two backward passes go to separate buffers, summed once (fixed order: deterministic).
        # Measured (scratch/ab_split.py, one process, one box): everything on the main stream 31.68-31.75 ms per step; the merged forward as a branch
        # beside the UNet head 31.93-32.00 (+0.25); split with the four smaller passes serial 33.6-33.7 (+2.0: latency-bound launches do not get
        # cheaper with fewer rows), branch P 32.04-32.16, both branches 31.97-32.07.  A latency-bound side chain hides only ~55 % of itself under
        # the one-round UNet kernels (a side workgroup delays the tile that wanted its CU by its whole duration) and every fork / join of the graph
        # costs a cross-stream dependency: both schedules are net LOSSES and stay opt-in (TB_SPLIT_TE=1, TB_TE_FWD_SIDE=1) as tested A/B knobs.
        self.split_te = self.merge_teacher and os.environ.get("TB_SPLIT_TE", "0") == "1"
        self.te_fwd_side = self.merge_teacher and os.environ.get("TB_TE_FWD_SIDE", "0") == "1"
        # Issue ORDER only, one stream: the encoder forward is issued where the UNet first needs the text states (in front of the hoisted K/V
        # projection), behind conv_in / the first ResNet block / the first self-attention.  Every replay opens with a submission bubble
        # (profiles/r05_step_timeline.txt: ~90 ... 250 us of gaps about 16 launches in, while the runtime is still writing the rest of the graph's
        # packets): ~0.7 ms of chip-filling kernels at the head of the graph cover it where 16 five-microsecond launches did not.  Same kernels,
        # same operands: bit-identical results.
        self.te_fwd_late = os.environ.get("TB_TE_FWD_LATE", "1") == "1"
        if self.merge_teacher:  # (allocated whenever the split is possible, so that tests can toggle `split_te` on one object)
            prio = int(os.environ.get("TB_SIDE_PRIORITY", "0"))
            self.side2 = torch.cuda.Stream(device=device, priority=prio)
            self.side = torch.cuda.Stream(device=device, priority=prio)
            self.flat_grad_p = torch.zeros(nA + nB + nE, device=device)
            self.grads_p = ((self.flat_grad_p[:nA].view_as(te.lora_A), self.flat_grad_p[nA:nA + nB].view_as(te.lora_B)) if has_lora else (None, None)) \
                + (self.flat_grad_p[nA + nB:].view(te.n_added, D),)
        self.vae = None  # attach_vae(): the step then starts from pixels (:1027-1037) instead of latents
        self.fused_tail = os.environ.get("TB_OPT_TAIL", "1") == "1"   # the optimizer tail as two launches (round 6) instead of ten
        self.opt_ws = torch.zeros(132, device=device)
        self.lr_table = None  # set_lr_table(): lambda(k) of --lr_scheduler on the device, indexed by the successful-step count

    def attach_vae(self, vae):
        """Run `vae.encode(pixel_values).latent_dist.sample() * scaling_factor` (:1036-1037) at the top of every step, inside the
        captured graph: `pixel_values` [B,3,8h,8w] fp32 becomes the step's static input and `x0` an internal buffer."""
        B, C, h, w = self.x0.shape
        assert vae.B == B and vae.H == 8 * h and vae.W == 8 * w and vae.geo.latent_channels == C
        self.vae = vae
        self.pixel_values = torch.zeros(B, vae.geo.in_channels, vae.H, vae.W, device=self.dev)
        self.vae_eps = torch.zeros(B, C, h, w, device=self.dev)

    # ------------------------------------------------------------------ pieces
    def draw(self):
        """:1041-1048 -- noise ~ N(0,1), timesteps ~ U{0..T-1} from torch generators (never inside custom kernels)."""
        if self.external_noise:
            return
        if self.vae is not None:  # DiagonalGaussianDistribution.sample() draws first (:1036), then the diffusion noise (:1041)
            self.vae_eps.normal_(generator=self.gen)
        self.noise.normal_(generator=self.gen)
        self.timesteps.random_(0, self.hp.num_train_timesteps, generator=self.gen)

    # the step body in four phases; `forward_backward` runs them in the reference's order on one stream (teacher on a forked side stream)
    def _phase_student(self, encoder_only=False, skip_encoder=False):
        hp, te, B = self.hp, self.te, self.B
        if not encoder_only:
            if self.vae is not None:
                self.x0.copy_(self.vae.encode(self.pixel_values, noise=self.vae_eps))              # :1027-1037
            ops.add_noise(self.x0, self.noise, self.timesteps, self.acp, self.noisy, self.velocity)
            te.pack_lora()
            self.unet.pack_kv_lora()
        if skip_encoder:
            return
        if self.merge_teacher:
            nb = self.ids_all.shape[0]
            out = te.forward(self.ids_all, slot=0, extra_ids=self.prior_ids, extra_table=self.teacher_table32)
            self.h_all, self.h_teacher = out[:nb * te.T], out[nb * te.T:]
        else:
            self.h_all = te.forward(self.ids_all, slot=0)                          # :1054-1059 and :1099 in one batch
        ops.convert(self.h_all[:B * te.T], self.ehs16)                             # .to(unet.dtype) :1066

    def _phase_teacher(self):
        """:1096-1106 -- the frozen fp16 teacher + the KPL loss; depends only on the student's hidden states."""
        hp, st = self.hp, self.state
        h0 = self.h_teacher if self.merge_teacher else self.teacher.forward(self.prior_ids, slot=0)
        kpl = ops.kpl_cos if hp.kpl_type == "cos" else ops.kpl_mse
        kpl(self.h_all[self.B * self.te.T:], h0, self.d_prior, self.kpl_partial, st[L.ST_LOSS_KPL:], st[L.ST_LOSS_SCALE:], hp.kpl_weight)

    def _phase_unet_forward(self):
        self.pred = self.unet.forward(self.noisy, self.timesteps, self.ehs16)      # :1063-1067

    def _phase_unet_backward(self):
        hp, st = self.hp, self.state
        if self.accum == 1:  # (accumulating: the buffer is zeroed by `replay` / `step_eager` at the start of a cycle instead)
            self.flat_grad.zero_()  # every trainable gradient is accumulated into this buffer (UNet adapters here, the encoder's below)
        target = self.noise if hp.prediction_type == "epsilon" else self.velocity  # :1070-1075
        ops.mse_loss(self.pred, target, self.dpred, st[L.ST_LOSS_MSE:], st[L.ST_LOSS_SCALE:])  # :1085-1090
        self.unet.backward(self.dpred, d_ehs_out=self.d_ehs)                       # :1108 (UNet part, dgrad only)

    def _phase_encoder_backward(self):
        hp, te = self.hp, self.te
        te.backward(self.d_all, slot=0)
        if hp.mixing is not None and te.r:  # :1119-1126 -- rows of each adapter's lora_B [D, r]: odd rows (object) / even rows (style) get no update
            gB = te.grad_B.view(te.geo.num_layers, 3, te.geo.hidden_size, te.r)
            gB[:, :, (1 if hp.mixing == "object" else 0)::2, :].zero_()

    def _forward_backward_split(self):
        """the step body with the text encoder split over three branches (see `split_te` in __init__)"""
        hp, te, B, st = self.hp, self.te, self.B, self.state
        BT = B * te.T
        main = torch.cuda.current_stream()
        if self.vae is not None:
            self.x0.copy_(self.vae.encode(self.pixel_values, noise=self.vae_eps))                  # :1027-1037
        ops.add_noise(self.x0, self.noise, self.timesteps, self.acp, self.noisy, self.velocity)
        te.pack_lora()
        self.unet.pack_kv_lora()
        fork = torch.cuda.Event()
        fork.record(main)
        # branch P: prior prompts through the trainable encoder (:1099) + the frozen teacher on the same prompts (:1096-1100), the KPL loss
        # (:1101-1106) and the backward of its term -- nothing here touches the UNet
        conc = getattr(self, "split_conc", 3)   # A/B knob: bit 0 = branch P on its own stream, bit 1 = branch I on its own stream
        sP = self.side if conc & 1 else main
        sI = self.side2 if conc & 2 else main
        sP.wait_event(fork)
        with torch.cuda.stream(sP), ops.workspace_slot(1):
            out = te.forward(self.prior_ids, slot=1, extra_ids=self.prior_ids, extra_table=self.teacher_table32)
            self.h_prior, self.h_teacher = out[:BT], out[BT:]
            kpl = ops.kpl_cos if hp.kpl_type == "cos" else ops.kpl_mse
            kpl(self.h_prior, self.h_teacher, self.d_prior, self.kpl_partial, st[L.ST_LOSS_KPL:], st[L.ST_LOSS_SCALE:], hp.kpl_weight)
            self.flat_grad_p.zero_()
            te.backward(self.d_prior, slot=1, grads=self.grads_p)
        # branch I: instance prompts -> encoder_hidden_states (:1054-1059), beside the UNet layers in front of the first cross-attention
        sI.wait_event(fork)
        with torch.cuda.stream(sI), ops.workspace_slot(2):
            self.h_inst = te.forward(self.input_ids, slot=0)
            ops.convert(self.h_inst, self.ehs16)                                                  # .to(unet.dtype) :1066
        self.pred = self.unet.forward(self.noisy, self.timesteps, self.ehs16, ehs_ready=lambda: main.wait_stream(sI))
        main.wait_stream(sI)
        self._phase_unet_backward()
        te.backward(self.d_ehs, slot=0)
        main.wait_stream(sP)
        n = self.flat_grad_p.numel()
        ops.add_f16(self.flat_grad[:n].view(1, n), self.flat_grad_p.view(1, n), self.flat_grad[:n].view(1, n))
        if hp.mixing is not None and te.r:  # :1119-1126
            gB = te.grad_B.view(te.geo.num_layers, 3, te.geo.hidden_size, te.r)
            gB[:, :, (1 if hp.mixing == "object" else 0)::2, :].zero_()

    def forward_backward(self):
        if self.split_te:
            return self._forward_backward_split()
        main = torch.cuda.current_stream()
        if self.te_fwd_side:
            # the encoder forward as a graph branch beside the UNet layers in front of the first cross-attention (conv_in, the first ResNet block,
            # the first self-attention: ~0.75 ms of chip-filling kernels next to ~90 latency-bound launches); joined by `ehs_ready`
            if self.vae is not None:
                self.x0.copy_(self.vae.encode(self.pixel_values, noise=self.vae_eps))              # :1027-1037
            ops.add_noise(self.x0, self.noise, self.timesteps, self.acp, self.noisy, self.velocity)
            self.te.pack_lora()
            self.unet.pack_kv_lora()
            self.side2.wait_stream(main)
            with torch.cuda.stream(self.side2), ops.workspace_slot(2):
                self._phase_student(encoder_only=True)
            self.pred = self.unet.forward(self.noisy, self.timesteps, self.ehs16, ehs_ready=lambda: main.wait_stream(self.side2))
            main.wait_stream(self.side2)
            self._phase_teacher()
            self._phase_unet_backward()
            self._phase_encoder_backward()
            return
        late = self.te_fwd_late and not (self.kpl and not self.merge_teacher)
        if late:
            self._phase_student(encoder_only=False, skip_encoder=True)
        else:
            self._phase_student()
        fork = None
        if self.kpl:
            fork = torch.cuda.Event()
            fork.record(main)
        if late:
            self.pred = self.unet.forward(self.noisy, self.timesteps, self.ehs16, ehs_ready=lambda: self._phase_student(encoder_only=True))
        else:
            self._phase_unet_forward()
        if self.kpl and self.merge_teacher:
            self._phase_teacher()  # only the KPL loss kernel is left of it
        elif self.kpl:
            # the teacher is issued on a side stream (a fork/join inside the HIP graph) after the UNet forward.  Branches of a graph do run
            # concurrently on ROCm 7 (scratch/graph_branches.py), but this one never paid (26.0 vs 26.0 / 26.1 steps/s in round 1, also as its
            # own graph on the side stream): a latency-bound side chain hides only about half of itself under the one-round UNet kernels
            # ("round 3, second half").
            self.side.wait_event(fork)
            with torch.cuda.stream(self.side), ops.workspace_slot(1):
                self._phase_teacher()
        self._phase_unet_backward()
        if self.kpl and not self.merge_teacher:
            main.wait_stream(self.side)
        self._phase_encoder_backward()

    def set_lr_multiplier(self, mult: float):
        """LambdaLR semantics of diffusers `get_scheduler` (:911-916, `lr_scheduler.step()` :1135): every param group's lr of the NEXT
        optimizer step is its base lr times `mult`.  A one-element device write outside the captured graph (no sync)."""
        self.state[L.ST_LR_MULT:L.ST_LR_MULT + 1].fill_(float(mult) - 1.0)  # the slot holds lambda - 1 (zeroed state = constant)

    def set_lr_table(self, lambdas):
        """diffusers get_scheduler + accelerate's AcceleratedScheduler (:911-916, :1135): lambdas[k] multiplies every group's base lr on the
        optimizer step that follows k successful (non-skipped) ones.  Looked up on the device inside the captured step: no host sync."""
        self.lr_table = torch.tensor([float(x) for x in lambdas], dtype=torch.float32, device=self.dev)

    def all_reduce(self):
        """DDP gradient averaging (:919-926): ONE RCCL all-reduce of the flat trainable-gradient buffer
        (k*D + 2*L*3*r*D floats ~ 0.94 MB at SD1.5, r=4) instead of the reference's dense 152.7 MB."""
        sum_gradients(self.flat_grad, self.world, force=self.force_dist)

    def optimizer_step(self):
        hp, te, st = self.hp, self.te, self.state
        D = te.geo.hidden_size
        if self.fused_tail:
            # the ten launches below as two (tb_optimizer_tail: same arithmetic in the same order, bit-equal -- test_gpu_edge.py); TB_OPT_TAIL=0 keeps them
            ne = self.n_emb
            ops.optimizer_tail(
                st, self.flat_grad, self.opt_ws,
                lora=(self.flat_lora, self.m_lora, self.v_lora) if self.n_lora else None,
                added=(te.token_table[te.first_added:], self.m_emb, self.v_emb) if te.n_added else None,
                unet=(self.flat_unet, self.m_unet, self.v_unet) if self.n_unet else None,
                decay=te.token_table[: te.first_added].view(-1), added_norms=self.added_norms if te.n_added else None,
                lr_table=self.lr_table, lr=hp.lr, emb_lr=hp.emb_lr, beta1=hp.beta1, beta2=hp.beta2, eps=hp.eps, wd=hp.wd,
                max_norm=hp.max_grad_norm, mean_norm=self.mean_norm, growth_interval=hp.growth_interval, use_scaler=hp.use_grad_scaler,
                grad_div=float(self.world * self.accum))
            return
        if self.n_lora:
            ops.sumsq(self.flat_grad[: self.n_lora], st[L.ST_SUMSQ_LORA:])
        if te.n_added:
            ops.sumsq(self.flat_grad[self.n_lora:self.n_lora + self.n_emb], st[L.ST_SUMSQ_EMB:])
        if self.lr_table is not None:
            ops.lr_from_table(st, self.lr_table)
        ops.scaler_update(st, hp.max_grad_norm, hp.beta1, hp.beta2, 2.0, 0.5, hp.growth_interval, hp.use_grad_scaler,
                          grad_div=float(self.world * self.accum))  # DDP's mean / accelerate's loss / gradient_accumulation_steps, folded
        if self.n_lora:
            ops.adamw(self.flat_lora, self.flat_grad[: self.n_lora], self.m_lora, self.v_lora, hp.lr, st, L.ST_COEF_LORA, hp.beta1,
                      hp.beta2, hp.eps, hp.wd)
        # group 0: the whole embedding matrix is an AdamW param; rows < first_added have zero grad (:1114-1117) and
        # therefore only see the decoupled decay
        orig = te.token_table[: te.first_added].view(-1)
        ops.weight_decay(orig, 1.0 - hp.emb_lr * hp.wd, st)
        if te.n_added:
            added = te.token_table[te.first_added:]
            ops.adamw(added.view(-1), self.flat_grad[self.n_lora:self.n_lora + self.n_emb], self.m_emb, self.v_emb, hp.emb_lr, st,
                      L.ST_COEF_EMB, hp.beta1, hp.beta2, hp.eps, hp.wd)
            ops.renorm_rows(added, self.mean_norm, self.added_norms)                # :1138-1149
        if self.n_unet:  # group 2 (:838-841): the unclipped unscale coefficient, the default learning rate
            ops.adamw(self.flat_unet, self.flat_grad[self.n_lora + self.n_emb:], self.m_unet, self.v_unet, hp.lr, st, L.ST_COEF_EMB,
                      hp.beta1, hp.beta2, hp.eps, hp.wd)

    def step_eager(self):
        """one loop iteration of the reference (:1024-1150).  With --gradient_accumulation_steps G > 1 an iteration is a MICRO step: gradients of G
        consecutive batches accumulate (accelerate divides each loss by G: folded into the unscale coefficient), the optimizer, the lr
        schedule and the step counter move on the G-th one only.  Returns True when the optimizer stepped (accelerator.sync_gradients)."""
        if self._micro == 0 and self.accum > 1:
            self.flat_grad.zero_()
        self.draw()
        self.forward_backward()
        self._micro += 1
        if self._micro < self.accum:
            return False
        self._micro = 0
        self.all_reduce()
        self.optimizer_step()
        return True

    # ------------------------------------------------------------------ HIP graph
    def capture(self, warmup: int = 2, single_graph: bool = True):
        """Capture draw + forward/backward [+ all-reduce] + optimizer into one HIP graph (static shapes, no host sync).
        Warm-up iterations run eagerly first (they DO update parameters, like any training step)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step_eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        dist = self.world > 1 or self.force_dist
        if self.accum > 1:  # two graphs: the micro step (draw + forward / backward, accumulating) and the optimizer tail of every G-th one
            self._micro = 0
            self.flat_grad.zero_()
            self.g1, self.g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g1, capture_error_mode="thread_local"):
                self.draw()
                self.forward_backward()
            with torch.cuda.graph(self.g2, capture_error_mode="thread_local"):
                self.optimizer_step()
            self.flat_grad.zero_()
            self.graph = (self.g1, self.g2)
            self.graph_mode = "micro+tail"
            return
        G = lambda: torch.cuda.CUDAGraph()  # noqa: E731
        cap = lambda g, **kw: torch.cuda.graph(g, capture_error_mode="thread_local", **kw)  # noqa: E731  (thread_local: the RCCL watchdog
        #                                                                      thread keeps polling its events while this thread captures)
        if dist and single_graph:
            import torch.distributed as tdist
            # only RCCL collectives can be captured; with any other backend (gloo in the tests) a failed capture attempt would also leave
            # torch's CUDA generator registered to the dead graph, so it is not attempted
            single_graph = tdist.is_initialized() and tdist.get_backend() == "nccl"
        if dist and single_graph:
            # RCCL collectives are stream-ordered and capturable: the whole step, exchange included, is ONE graph (no host involvement
            # between backward and optimizer).  Falls back to two graphs around an eager all-reduce if the capture is refused.
            gens = [torch.cuda.default_generators[torch.cuda.current_device()]] + ([self.gen] if self.gen is not None else [])
            snaps = [g_.get_state() for g_ in gens]
            try:
                g = G()
                with cap(g):
                    self.draw()
                    self.forward_backward()
                    self.all_reduce()
                    self.optimizer_step()
                self.graph = (g,)
                self.graph_mode = "single+rccl"
                return
            except Exception as e:  # noqa: BLE001 -- capture refused by this RCCL / torch build
                import warnings
                warnings.warn(f"RCCL all-reduce could not be captured in the step graph ({e}); using two graphs around an eager collective")
                g = None
                torch.cuda.synchronize()
                # a capture that died leaves torch's CUDA generator states flagged "capturing" (capture_end never ran their epilogue) and
                # the next capture_begin refuses to register them: swap in fresh state objects carrying the pre-capture seed / offset
                for g_, snap in zip(gens, snaps):
                    fresh = torch.Generator(device=g_.device)
                    fresh.set_state(snap)
                    g_.graphsafe_set_state(fresh.graphsafe_get_state())
        if dist:
            # the collective outside the graphs: two graphs around one eager RCCL call
            self.g1, self.g2 = G(), G()
            with cap(self.g1):
                self.draw()
                self.forward_backward()
            with cap(self.g2):
                self.optimizer_step()
            self.graph = (self.g1, self.g2)
            self.graph_mode = "two+eager-rccl"
        else:
            g = G()
            with cap(g):
                self.draw()
                self.forward_backward()
                self.optimizer_step()
            self.graph = (g,)
            self.graph_mode = "single"

    def replay(self):
        if self.graph is None:
            return self.step_eager()
        if self.accum > 1:
            if self._micro == 0:
                self.flat_grad.zero_()
            self.graph[0].replay()
            self._micro += 1
            if self._micro < self.accum:
                return False
            self._micro = 0
            self.graph[1].replay()
            return True
        if len(self.graph) == 1:
            self.graph[0].replay()
        else:
            self.graph[0].replay()
            self.all_reduce()
            self.graph[1].replay()
        return True

    # ------------------------------------------------------------------ host-side reads (NOT in the hot loop)
    def scalars(self):
        s = self.state.tolist()
        return {"loss_mse": s[L.ST_LOSS_MSE], "loss_kpl": s[L.ST_LOSS_KPL],
                "loss": s[L.ST_LOSS_MSE] + self.hp.kpl_weight * s[L.ST_LOSS_KPL], "loss_scale": s[L.ST_LOSS_SCALE],
                "grad_norm": s[L.ST_GRAD_NORM], "found_inf": s[L.ST_FOUND_INF], "opt_steps": s[L.ST_STEP]}
