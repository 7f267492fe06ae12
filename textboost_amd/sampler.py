"""Validation / inference sampling on MI355X -- the device work of the reference's `log_validation` (train_textboost.py:453-531) and
`inference.py:73-100`: diffusers `StableDiffusionPipeline.__call__` with `DPMSolverMultistepScheduler`, 25 steps, guidance 7.5.
SURVEY.md 8(f) row 2.

    ids (prompt, empty prompt) -> text encoder -> [classifier-free guidance: UNet on (x, x) with (uncond, cond)] x steps
    -> DPM-Solver++(2M) update -> VAE decoder -> image in [0, 1]

The UNet forward, the text encoder and the VAE decoder are the executors of this package; the per-step scalar coefficients of the solver
are host arithmetic (`DPMSolverPP2M`, float64, computed once per schedule), the update itself is one elementwise kernel (`tb_dpm_step`).
Nothing here synchronises with the host, so the whole loop can be captured in a HIP graph.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch

from . import ops
from . import _lib as L


class DPMSolverPP2M:
    """diffusers DPMSolverMultistepScheduler as `from_config(pipeline.scheduler.config)` builds it (train_textboost.py:493-495): algorithm
    "dpmsolver++", solver_order 2, solver_type "midpoint", final sigma zero (last step first order), and -- inherited from the model's own
    scheduler config (`<model>/scheduler/scheduler_config.json`; a PNDMScheduler for SD1.x / SD2.1-base, a DDIMScheduler for SD2.1-768) --
    the betas, `prediction_type` ("epsilon" | "v_prediction"), `timestep_spacing` and `steps_offset`.  PNDM / DDIM instances carry
    timestep_spacing "leading" (their class default) and SD's files say steps_offset 1, so that is the default when no file exists.
    sigma = sqrt((1 - abar) / abar); alpha_t = 1 / sqrt(sigma^2 + 1); sigma_t = sigma alpha_t; lambda = log(alpha_t / sigma_t).
    [3P: diffusers 0.29 is not installed here; the three spacings are restated from its set_timesteps.]"""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 prediction_type="epsilon", timestep_spacing="leading", steps_offset=1):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type {prediction_type}")
        if timestep_spacing not in ("linspace", "leading", "trailing"):
            raise ValueError(f"timestep_spacing {timestep_spacing}")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        self.T = num_train_timesteps
        self.prediction_type, self.timestep_spacing, self.steps_offset = prediction_type, timestep_spacing, int(steps_offset)
        self.init_noise_sigma = 1.0

    @classmethod
    def from_config(cls, cfg: Optional[dict]):
        """cfg = the parsed scheduler_config.json of the model (None: SD1.x's values)."""
        cfg = cfg or {}
        return cls(num_train_timesteps=cfg.get("num_train_timesteps", 1000), beta_start=cfg.get("beta_start", 0.00085),
                   beta_end=cfg.get("beta_end", 0.012), beta_schedule=cfg.get("beta_schedule", "scaled_linear"),
                   prediction_type=cfg.get("prediction_type", "epsilon"), timestep_spacing=cfg.get("timestep_spacing", "leading"),
                   steps_offset=cfg.get("steps_offset", 1))

    def set_timesteps(self, n: int) -> List[int]:
        T = self.T  # last_timestep (lambda_min_clipped = -inf: nothing is clipped)
        if self.timestep_spacing == "linspace":
            ts = torch.linspace(0, T - 1, n + 1, dtype=torch.float64).round().flip(0)[:-1].long()
        elif self.timestep_spacing == "leading":
            ratio = T // (n + 1)
            ts = (torch.arange(0, n + 1, dtype=torch.float64) * ratio).round().flip(0)[:-1].long() + self.steps_offset
        else:  # trailing
            ts = torch.arange(T, 0, -T / n, dtype=torch.float64).round().long() - 1
        ac = self.alphas_cumprod[ts]
        self.timesteps = ts.tolist()
        self.sigmas = ((1 - ac) / ac).sqrt().tolist() + [0.0]
        return self.timesteps

    @staticmethod
    def alpha_sigma(sigma: float) -> Tuple[float, float]:
        a = 1.0 / math.sqrt(sigma * sigma + 1.0)
        return a, sigma * a

    def data_prediction_scalars(self, i: int) -> Tuple[float, float]:
        """(a, s) such that the data prediction of step i is m0 = (x - s * model_output) / a -- the form `tb_dpm_step` evaluates.
        epsilon: m0 = (x - sigma_t eps) / alpha_t;  v_prediction: m0 = alpha_t x - sigma_t v = (x - (sigma_t / alpha_t) v) / (1 / alpha_t)."""
        a_t, s_t = self.alpha_sigma(self.sigmas[i])
        return (a_t, s_t) if self.prediction_type == "epsilon" else (1.0 / a_t, s_t / a_t)

    def coefficients(self, i: int) -> Tuple[float, float, float]:
        """x_next = ca x + cb m0 + cc m_prev  for step i  (m = data prediction)."""
        a0, st0 = self.alpha_sigma(self.sigmas[i])
        a1, st1 = self.alpha_sigma(self.sigmas[i + 1])
        if i == len(self.timesteps) - 1:      # sigma_next = 0: exp(-h) = 0, alpha_next = 1
            return 0.0, a1, 0.0
        lam0, lam1 = math.log(a0 / st0), math.log(a1 / st1)
        h = lam1 - lam0
        e = math.expm1(-h)
        ca = st1 / st0
        if i == 0:
            return ca, -a1 * e, 0.0
        ap, stp = self.alpha_sigma(self.sigmas[i - 1])
        r0 = (lam0 - math.log(ap / stp)) / h
        return ca, -a1 * e * (1.0 + 0.5 / r0), a1 * e * (0.5 / r0)


class DDPMAncestral:
    """--validation_scheduler DDPMScheduler (train_textboost.py:341-345, :483-495): diffusers DDPMScheduler built from the model's scheduler config
    with variance_type "fixed_small" -- SD's betas, clip_sample false, timestep_spacing "leading" (timesteps arange(n) * (T // n), reversed,
    + steps_offset), the ancestral step  x_prev = c_x0 x0 + c_x x + sqrt(var) z.  In `tb_dpm_step`'s form (x = ca x + cb m0 + cc m_prev, m0 the
    data prediction) the variance noise z rides in the m_prev operand: (ca, cb, cc) = (c_x, c_x0, sqrt(var)).  [3P restated]"""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", prediction_type="epsilon",
                 steps_offset=1):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type {prediction_type}")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        self.T, self.prediction_type, self.steps_offset = num_train_timesteps, prediction_type, int(steps_offset)
        self.init_noise_sigma = 1.0
        self.needs_noise = True

    @classmethod
    def from_config(cls, cfg: Optional[dict]):
        cfg = cfg or {}
        if cfg.get("clip_sample", False):
            raise NotImplementedError("DDPMScheduler with clip_sample = true (SD's scheduler configs say false)")
        return cls(num_train_timesteps=cfg.get("num_train_timesteps", 1000), beta_start=cfg.get("beta_start", 0.00085),
                   beta_end=cfg.get("beta_end", 0.012), beta_schedule=cfg.get("beta_schedule", "scaled_linear"),
                   prediction_type=cfg.get("prediction_type", "epsilon"), steps_offset=cfg.get("steps_offset", 1))

    def set_timesteps(self, n: int) -> List[int]:
        self.n = n
        self.timesteps = ((torch.arange(0, n, dtype=torch.float64) * (self.T // n)).round().flip(0).long() + self.steps_offset).tolist()
        return self.timesteps

    def data_prediction_scalars(self, i: int) -> Tuple[float, float]:
        ab = float(self.alphas_cumprod[self.timesteps[i]])
        a_t, s_t = math.sqrt(ab), math.sqrt(1.0 - ab)
        return (a_t, s_t) if self.prediction_type == "epsilon" else (1.0 / a_t, s_t / a_t)

    def coefficients(self, i: int) -> Tuple[float, float, float]:
        t = self.timesteps[i]
        prev_t = t - self.T // self.n
        ab_t = float(self.alphas_cumprod[t])
        ab_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        alpha = ab_t / ab_p
        beta = 1.0 - alpha
        var = max((1.0 - ab_p) / (1.0 - ab_t) * beta, 1e-20)
        return math.sqrt(alpha) * (1.0 - ab_p) / (1.0 - ab_t), math.sqrt(ab_p) * beta / (1.0 - ab_t), (math.sqrt(var) if t > 0 else 0.0)


class HipSampler:
    """`sample(cond_ehs, uncond_ehs, latents=None) -> images [B,3,8h,8w] in [0,1]`.  `unet` must be built for batch 2B (guidance runs the
    unconditional and conditional rows in one call, as the pipeline does), `vae_decoder` for batch B."""

    def __init__(self, unet, vae_decoder, steps: int = 25, guidance: float = 7.5, scheduler_config: Optional[dict] = None,
                 scheduler: str = "DPMSolverMultistepScheduler"):
        assert unet.B == 2 * vae_decoder.B and unet.H == vae_decoder.h and unet.W == vae_decoder.w
        self.unet, self.vae, self.steps, self.g = unet, vae_decoder, steps, guidance
        self.B = vae_decoder.B
        if scheduler == "DPMSolverMultistepScheduler":
            self.sch = DPMSolverPP2M.from_config(scheduler_config)
        elif scheduler == "DDPMScheduler":  # --validation_scheduler DDPMScheduler (:341-345)
            self.sch = DDPMAncestral.from_config(scheduler_config)
        else:
            raise ValueError(f"validation scheduler {scheduler}")
        self.timesteps = self.sch.set_timesteps(steps)
        dev = unet.dev
        B, h, w = self.B, unet.H, unet.W
        self.x = torch.zeros(B, 4, h, w, device=dev)
        self.m_prev = torch.zeros(B, 4, h, w, device=dev)
        self.x2 = torch.zeros(2 * B, 4, h, w, device=dev, dtype=L.half_dtype())
        self.t_dev = [torch.full((2 * B,), t, dtype=torch.int64, device=dev) for t in self.timesteps]
        self.generator: Optional[torch.Generator] = None

    def denoise(self, cond_ehs, uncond_ehs, latents=None, step_noise=None):
        """cond / uncond: fp16 or fp32 [B*77, D] text-encoder outputs.  Returns the final latents fp32 [B,4,h,w].
        step_noise (ancestral schedulers only): one [B,4,h,w] tensor per step instead of drawing from `self.generator`."""
        B, sch = self.B, self.sch
        if latents is None:
            latents = torch.randn(self.x.shape, device=self.x.device, generator=self.generator)
        self.x.copy_(latents * sch.init_noise_sigma)
        self.m_prev.zero_()
        self.x2[:B].copy_(self.x)
        self.x2[B:].copy_(self.x)
        ehs = torch.cat([uncond_ehs, cond_ehs]).to(L.half_dtype()).contiguous()
        n = self.x[0].numel()
        for i in range(self.steps):
            eps2 = self.unet.forward(self.x2, self.t_dev[i], ehs)
            a_t, s_t = sch.data_prediction_scalars(i)
            ca, cb, cc = sch.coefficients(i)
            if getattr(sch, "needs_noise", False):  # the variance noise of the ancestral step rides in the m_prev operand
                if step_noise is not None:
                    self.m_prev.copy_(step_noise[i])
                else:
                    self.m_prev.normal_(generator=self.generator)
            ops.dpm_step(self.x, eps2, self.m_prev, self.x2, n, B, self.g, a_t, s_t, ca, cb, cc)
        return self.x

    def sample(self, cond_ehs, uncond_ehs, latents=None):
        return self.vae.decode(self.denoise(cond_ehs, uncond_ehs, latents))


def make_image_grid(images: torch.Tensor, rows: int, cols: int):
    """diffusers.utils.make_image_grid on a [rows*cols, 3, H, W] tensor in [0, 1] -> one PIL image (train_textboost.py:1224-1228)."""
    from PIL import Image
    n, c, H, W = images.shape
    assert n == rows * cols and c == 3
    arr = (images.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
    grid = Image.new("RGB", (cols * W, rows * H))
    for i in range(n):
        grid.paste(Image.fromarray(arr[i]), box=((i % cols) * W, (i // cols) * H))
    return grid
