"""Synthetic TextBoost workload of SURVEY.md 8(d) / BASELINE.json configs: seeded random-init SD1.5-shaped models,
synthetic 4x64x64 latents and CLIP token ids (no image files, no VAE, no checkpoints -- there is no network here)."""
from __future__ import annotations

import torch

from . import models
from .text_encoder import CLIPGeometry, HipTextEncoder
from .trainer import StepHyper, TextBoostStep
from .unet import HipUNet, UNetGeometry

BOS, EOS = 49406, 49407


def synthetic_ids(B, added_ids, gen: torch.Generator, prior=False, null_prob=0.1, vocab=49406, T=77):
    """instance prompts: BOS, 3..12 random tokens with one placeholder (and w.p. 0.5 two augmentation tokens), EOS padding;
    prior prompts: no added ids, 10 % null prompts (--null_prob, train_textboost.py:412-417)."""
    ids = torch.full((B, T), EOS, dtype=torch.int64)
    ids[:, 0] = BOS
    for b in range(B):
        if prior and torch.rand((), generator=gen).item() < null_prob:
            continue
        n = int(torch.randint(3, 13, (), generator=gen))
        ids[b, 1:1 + n] = torch.randint(0, vocab, (n,), generator=gen)
        if not prior:
            pos = 1 + int(torch.randint(0, n, (), generator=gen))
            ids[b, pos] = added_ids[0]
            if len(added_ids) > 2 and torch.rand((), generator=gen).item() < 0.5:
                others = [p for p in range(1, 1 + n) if p != pos][:2]
                for j, p in enumerate(others):
                    ids[b, p] = added_ids[1 + j]
    return ids


def build_step(batch=8, latent=64, unet_geo: UNetGeometry = models.SD15_UNET, clip_geo: CLIPGeometry = models.SD15_CLIP,
               lora_rank=4, n_added=18, hyper: StepHyper | None = None, weight_seed=1234, data_seed=1000, device="cuda",
               world_size=1, with_vae=False, attn_fp8=False, precision="fp16"):
    """Config 2 of BASELINE.json by default: SD1.5 UNet + CLIP-L, per-GPU batch 8, 512^2 (64^2 latents), LoRA r=4, KPL on,
    18 added token vectors (2 placeholder + 16 augmentation vectors).
    precision: "fp16" = the reference driver's --mixed_precision fp16 (run_textboost_db.py:150); "fp32" = its default no-AMP mode
    (train_textboost.py:298-308, the README command): fp32 everywhere, no GradScaler; "bf16" = --mixed_precision bf16 (:930-934: bf16 UNet /
    teacher, bf16 autocast operands in the trainable encoder, no GradScaler -- accelerate creates one for fp16 only): the process switches to the
    bfloat16 build of the library (`_lib.set_half("bf16")`)."""
    assert precision in ("fp16", "fp32", "bf16")
    from . import _lib
    _lib.set_half("fp16" if precision == "fp32" else precision)   # (fp32 mode: the 16-bit helpers -- VAE, sampler -- are the fp16 build's)
    f32 = precision == "fp32"
    hyper = hyper or (StepHyper(use_grad_scaler=False, init_scale=1.0) if precision != "fp16" else StepHyper())
    usd = models.random_state_dict(models.unet_shapes(unet_geo), weight_seed, device=device)
    unet = HipUNet(unet_geo, usd, batch, latent, latent, text_len=clip_geo.max_pos, device=device, attn_fp8=attn_fp8,
                   dtype=torch.float32 if f32 else _lib.half_dtype())
    del usd
    csd = models.random_state_dict(models.clip_shapes(clip_geo), weight_seed + 1, device=device)
    teacher = HipTextEncoder(clip_geo, csd, batch, mode="fp32" if f32 else "half", device=device)
    # null embedding = frozen encoder output for the empty prompt (the reference ships one only for SD2.1)
    null_ids = torch.full((1, clip_geo.max_pos), EOS, dtype=torch.int64, device=device)
    null_ids[0, 0] = BOS
    frozen = HipTextEncoder(clip_geo, csd, 1, mode="fp32" if f32 else "autocast", device=device)
    null = frozen.forward(null_ids, pins=False).clone()
    del frozen
    te = HipTextEncoder(clip_geo, csd, batch, mode="fp32" if f32 else "autocast", lora_rank=lora_rank, n_slots=2, device=device,
                        seed=weight_seed + 2)
    del csd
    te.set_null_embedding(null)
    teacher.set_null_embedding(null)
    g = torch.Generator().manual_seed(weight_seed + 3)
    init_ids = torch.randint(0, 49406, (n_added,), generator=g).tolist()
    added = te.add_tokens(init_ids)
    step = TextBoostStep(unet, te, teacher, hyper, (batch, 4, latent, latent), device=device, world_size=world_size)
    dg = torch.Generator().manual_seed(data_seed)
    step.x0.copy_(torch.randn(batch, 4, latent, latent, generator=dg))
    if with_vae:  # the reference's full step: pixels in [-1, 1] -> VAE encoder -> latents (train_textboost.py:1027-1037)
        from .vae import HipVAEEncoder, VAEGeometry, vae_encoder_shapes
        vgeo = VAEGeometry()
        vsd = models.random_state_dict(vae_encoder_shapes(vgeo), weight_seed + 4, device=device)
        step.attach_vae(HipVAEEncoder(vgeo, vsd, batch, 8 * latent, 8 * latent, device=device))
        step.pixel_values.copy_(torch.rand(batch, 3, 8 * latent, 8 * latent, generator=dg) * 2 - 1)
    step.input_ids.copy_(synthetic_ids(batch, added, dg))
    step.prior_ids.copy_(synthetic_ids(batch, added, dg, prior=True))
    return step, added
