"""ORACLE (test infrastructure only) -- eager-PyTorch fp32 restatement of the reference's validation / inference sampling path
(SURVEY.md 8(f) row 2):

    train_textboost.py:453-531 `log_validation`  -> diffusers `StableDiffusionPipeline(prompt, num_inference_steps=25)` with
                                                    `args.validation_scheduler` (default DPMSolverMultistepScheduler), guidance 7.5
    inference.py:73-100                           -> the same pipeline with `DPMSolverMultistepScheduler.from_config(...)`

Three third-party pieces (diffusers==0.29.0, pyproject.toml:12 -- NOT vendored under /root/reference, NOT installed here), restated from
their published algorithms:
  * `AutoencoderKL.decode`: `post_quant_conv` + `Decoder` (conv_in, UNetMidBlock2D with one single-head attention, 4 UpDecoderBlock2D of
    3 resnets (+ nearest-x2 Upsample2D with a 3x3 conv on the first three), GroupNorm/SiLU/conv_out);
  * `DPMSolverMultistepScheduler` with the defaults `from_config` fills in on top of SD1.x's scheduler config: dpmsolver++, solver_order 2,
    prediction type / timestep spacing ("leading", steps_offset 1) inherited from the model's scheduler config, lower_order_final / final
    sigma zero (Lu et al., DPM-Solver++ 2M, multistep midpoint);
  * the pipeline's classifier-free-guidance loop.

Pinning: "parity unpinned" (no reference tests, diffusers absent).  What IS pinned (tests/test_oracle_sampler.py): the published decoder
parameter count 49,490,179 (+ 20 for post_quant_conv; with the encoder's 34,163,592 + 72 = the SD VAE's 83,653,863), the closed-form
exactness of DPM-Solver++ on a model that predicts the true noise of a point mass (any consistent solver must land on the point), and
first-order / second-order coefficient identities.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .vae_encoder import AttnBlock, ResnetBlock2D, VAEConfig


class Upsample2D(nn.Module):
    """diffusers Upsample2D(use_conv=True): nearest x2 then a 3x3 conv."""

    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpDecoderBlock2D(nn.Module):
    def __init__(self, cfg: VAEConfig, cin, cout, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if j == 0 else cout, cout, cfg.norm_num_groups, cfg.norm_eps)
                                      for j in range(cfg.layers_per_block + 1)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Mid(nn.Module):
    def __init__(self, cfg: VAEConfig, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, cfg.norm_num_groups, cfg.norm_eps) for _ in range(2)])
        self.attentions = nn.ModuleList([AttnBlock(ch, cfg.norm_num_groups, cfg.norm_eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = list(reversed(cfg.block_out_channels))
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = _Mid(cfg, ch[0])
        self.up_blocks = nn.ModuleList()
        prev = ch[0]
        for i, c in enumerate(ch):
            self.up_blocks.append(UpDecoderBlock2D(cfg, prev, c, add_up=i < len(ch) - 1))
            prev = c
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[-1], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[-1], cfg.in_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VAEDecoder(nn.Module):
    """`AutoencoderKL.decode(z).sample`; `decode_latents` adds the pipeline's `1 / scaling_factor` and [0, 1] post-processing."""

    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = Decoder(cfg)

    def forward(self, z):
        return self.decoder(self.post_quant_conv(z))

    def decode_latents(self, latents):
        """StableDiffusionPipeline: image = vae.decode(latents / scaling_factor).sample; (image / 2 + 0.5).clamp(0, 1)."""
        return (self(latents / self.cfg.scaling_factor) / 2 + 0.5).clamp(0, 1)


def count_decoder_params(cfg: VAEConfig = VAEConfig()):
    with torch.device("meta"):
        m = VAEDecoder(cfg)
    return sum(p.numel() for p in m.decoder.parameters()), sum(p.numel() for p in m.post_quant_conv.parameters())


# ----------------------------------------------------------------------------------------------------------------- scheduler
class DPMSolverPP2M:
    """DPMSolverMultistepScheduler(algorithm_type="dpmsolver++", solver_order=2, solver_type="midpoint", prediction_type="epsilon",
    lower_order_final=True, final_sigmas_type="zero") on SD's scaled-linear betas; timestep_spacing / steps_offset / prediction_type as inherited.

    Notation of the diffusers implementation: sigma = sqrt((1 - abar) / abar), alpha_t = 1 / sqrt(sigma^2 + 1), sigma_t = sigma * alpha_t,
    lambda = log(alpha_t) - log(sigma_t).  Data prediction x0 = (x - sigma_t * eps) / alpha_t.
      first order : x_next = (sigma_next / sigma_cur) x - alpha_next (exp(-h) - 1) D0,                 h = lambda_next - lambda_cur
      second order: x_next = (sigma_next / sigma_cur) x - alpha_next (exp(-h) - 1) D0 - 0.5 alpha_next (exp(-h) - 1) D1,
                    D0 = m0, D1 = (m0 - m1) / r0, r0 = h_prev / h
    The first step and (final sigma zero) the last step are first order."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=1):
        """`from_config(pipeline.scheduler.config)` (train_textboost.py:493-495) inherits prediction_type / timestep_spacing / steps_offset
        from the model's PNDM (SD1.x, SD2.1-base) or DDIM (SD2.1-768) scheduler instance: spacing "leading" (their class default),
        steps_offset 1 (SD's scheduler_config.json) -- the defaults here."""
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.T = num_train_timesteps
        self.prediction_type, self.timestep_spacing, self.steps_offset = prediction_type, timestep_spacing, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        ac = self.alphas_cumprod.double()
        if self.timestep_spacing == "linspace":
            ts = torch.linspace(0, self.T - 1, n + 1, dtype=torch.float64).round().flip(0)[:-1].long()
        elif self.timestep_spacing == "leading":   # DPMSolverMultistepScheduler.set_timesteps: step_ratio = last_timestep // (n + 1)
            ts = (torch.arange(0, n + 1, dtype=torch.float64) * (self.T // (n + 1))).round().flip(0)[:-1].long() + self.steps_offset
        else:                                       # "trailing"
            ts = torch.arange(self.T, 0, -self.T / n, dtype=torch.float64).round().long() - 1
        sig_all = ((1 - ac) / ac).sqrt()
        self.timesteps = ts
        self.sigmas = torch.cat([sig_all[ts], torch.zeros(1, dtype=torch.float64)])  # final_sigmas_type = "zero"
        self.step_index = 0
        self.m_prev: Optional[torch.Tensor] = None
        return ts

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha_t, sigma * alpha_t

    def coefficients(self, i: int):
        """(a, b, c) with x_next = a x + b m0 + c m_prev for step i (c = 0 on first-order steps)."""
        s0, s1 = self.sigmas[i], self.sigmas[i + 1]
        a0, st0 = self._alpha_sigma(s0)
        a1, st1 = self._alpha_sigma(s1)
        last = i == len(self.timesteps) - 1
        if last:  # sigma_next = 0: lambda_next = +inf, exp(-h) = 0
            return 0.0, float(a1), 0.0  # x_next = alpha_next * m0 = m0 (alpha(0) = 1)
        lam0, lam1 = torch.log(a0) - torch.log(st0), torch.log(a1) - torch.log(st1)
        h = lam1 - lam0
        a = float(st1 / st0)
        e = float(torch.exp(-h) - 1.0)
        if i == 0:
            return a, float(-a1 * e), 0.0
        sp = self.sigmas[i - 1]
        ap, stp = self._alpha_sigma(sp)
        lamp = torch.log(ap) - torch.log(stp)
        r0 = float((lam0 - lamp) / h)
        # D0 = m0, D1 = (m0 - m_prev) / r0:  -a1 e m0 - 0.5 a1 e (m0 - m_prev)/r0
        b = float(-a1 * e) * (1.0 + 0.5 / r0)
        c = float(a1 * e) * (0.5 / r0)
        return a, b, c

    def step(self, eps, sample):
        i = self.step_index
        a_t, s_t = self._alpha_sigma(self.sigmas[i])
        if self.prediction_type == "epsilon":
            m0 = (sample - float(s_t) * eps) / float(a_t)
        else:  # v_prediction: x0 = alpha_t x - sigma_t v
            m0 = float(a_t) * sample - float(s_t) * eps
        a, b, c = self.coefficients(i)
        out = a * sample + b * m0 + (c * self.m_prev if c != 0.0 else 0.0)
        self.m_prev = m0
        self.step_index += 1
        return out


class DDPMAncestral:
    """diffusers 0.29 DDPMScheduler as `DDPMScheduler.from_config(pipeline.scheduler.config, variance_type="fixed_small")` builds it for
    --validation_scheduler DDPMScheduler (/root/reference/train_textboost.py:341-345, :483-495): SD's betas, clip_sample false (SD's config),
    timestep_spacing "leading" with steps_offset (timesteps = arange(n) * (T // n), reversed, + offset), ancestral step
        prev_t = t - T // n;  abar_prev = abar[prev_t] (1 for prev_t < 0);  alpha = abar_t / abar_prev;  beta = 1 - alpha
        x0 = (x - sqrt(1 - abar_t) eps) / sqrt(abar_t)      (v-prediction: sqrt(abar_t) x - sqrt(1 - abar_t) v)
        x_prev = sqrt(abar_prev) beta / (1 - abar_t) x0 + sqrt(alpha) (1 - abar_prev) / (1 - abar_t) x + sqrt(var) z   (z for t > 0),
        var = max((1 - abar_prev) / (1 - abar_t) beta, 1e-20).   [3P: restated, diffusers is not installed here]"""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon", steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0).double()
        self.T, self.prediction_type, self.steps_offset = num_train_timesteps, prediction_type, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        self.n = n
        self.timesteps = ((torch.arange(0, n, dtype=torch.float64) * (self.T // n)).round().flip(0).long() + self.steps_offset)
        return self.timesteps

    def coefficients(self, t: int):
        """(c_x0, c_x, sqrt(var)) of the step at timestep t"""
        prev_t = t - self.T // self.n
        ab_t = float(self.alphas_cumprod[t])
        ab_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        alpha = ab_t / ab_p
        beta = 1.0 - alpha
        c_x0 = ab_p ** 0.5 * beta / (1.0 - ab_t)
        c_x = alpha ** 0.5 * (1.0 - ab_p) / (1.0 - ab_t)
        var = max((1.0 - ab_p) / (1.0 - ab_t) * beta, 1e-20)
        return c_x0, c_x, (var ** 0.5 if t > 0 else 0.0)

    def step(self, eps, t: int, sample, noise):
        ab_t = float(self.alphas_cumprod[t])
        if self.prediction_type == "epsilon":
            x0 = (sample - (1.0 - ab_t) ** 0.5 * eps) / ab_t ** 0.5
        else:
            x0 = ab_t ** 0.5 * sample - (1.0 - ab_t) ** 0.5 * eps
        c_x0, c_x, sd = self.coefficients(t)
        return c_x0 * x0 + c_x * sample + sd * noise


def sample_latents_ddpm(unet: Callable, cond, uncond, latents, step_noise, steps=25, guidance=7.5, **scheduler_kwargs):
    """the guided loop with the ancestral DDPM step; step_noise[i] = the variance noise of step i (the pipeline draws it from its generator)"""
    sch = DDPMAncestral(**scheduler_kwargs)
    ts = sch.set_timesteps(steps)
    x = latents * sch.init_noise_sigma
    B = x.shape[0]
    ehs = torch.cat([uncond, cond])
    for i, t in enumerate(ts.tolist()):
        e = unet(torch.cat([x, x]), torch.full((2 * B,), t, dtype=torch.long), ehs)
        eps = e[:B] + guidance * (e[B:] - e[:B])
        x = sch.step(eps, t, x, step_noise[i])
    return x


def sample_latents(unet: Callable, cond, uncond, latents, steps=25, guidance=7.5, **scheduler_kwargs):
    """The denoising loop of StableDiffusionPipeline.__call__ with classifier-free guidance (do_classifier_free_guidance = g > 1):
    eps = eps_uncond + g (eps_cond - eps_uncond); `unet(x[2B], t[2B], ehs[2B])` -> eps[2B]."""
    sch = DPMSolverPP2M(**scheduler_kwargs)
    ts = sch.set_timesteps(steps)
    x = latents * sch.init_noise_sigma
    B = x.shape[0]
    ehs = torch.cat([uncond, cond])
    for t in ts.tolist():
        e = unet(torch.cat([x, x]), torch.full((2 * B,), t, dtype=torch.long), ehs)
        eps = e[:B] + guidance * (e[B:] - e[:B])
        x = sch.step(eps, x)
    return x
