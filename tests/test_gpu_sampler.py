"""Validation / inference sampling path (log_validation train_textboost.py:453-531, inference.py) on the GPU vs the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _decoder_pair(cfg_oracle, geo, seed, B, h, w):
    from oracle.sampler import VAEDecoder
    from textboost_amd import models
    from textboost_amd.vae import HipVAEDecoder, vae_decoder_shapes
    sd = models.random_state_dict(vae_decoder_shapes(geo), seed, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}
    with torch.device("meta"):
        ref = VAEDecoder(cfg_oracle)
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    return ref, HipVAEDecoder(geo, {k: v.to(dev) for k, v in sd.items()}, B, h, w, device=dev)


@pytest.mark.parametrize("spacing,offset,pred", [("leading", 1, "epsilon"), ("linspace", 0, "epsilon"), ("trailing", 0, "v_prediction"),
                                                 ("leading", 1, "v_prediction")])
def test_dpm_step_kernel_follows_the_oracle_scheduler(spacing, offset, pred):
    from oracle.sampler import DPMSolverPP2M as Ref
    from textboost_amd import ops
    from textboost_amd.sampler import DPMSolverPP2M
    ref = Ref(prediction_type=pred, timestep_spacing=spacing, steps_offset=offset)
    sch = DPMSolverPP2M.from_config({"prediction_type": pred, "timestep_spacing": spacing, "steps_offset": offset})
    ts = ref.set_timesteps(25)
    assert sch.set_timesteps(25) == ts.tolist()
    torch.manual_seed(0)
    B, n, g = 2, 4 * 16 * 16, 7.5
    x_ref = torch.randn(B, n)
    x = x_ref.clone().to(dev); m_prev = torch.zeros(B, n, device=dev); x2 = torch.zeros(2 * B, n, device=dev, dtype=torch.float16)
    for i in range(25):
        e = (torch.randn(2 * B, n) * 0.8).half()
        eps = e[:B].float() + g * (e[B:].float() - e[:B].float())
        x_ref = ref.step(eps, x_ref)
        a_t, s_t = sch.data_prediction_scalars(i)
        ops.dpm_step(x, e.to(dev), m_prev, x2, n, B, g, a_t, s_t, *sch.coefficients(i))
        assert rel_err(x, x_ref) < 1e-5, i
        # x2 = fp16 of the update (the compiler may round the exact fma once, v_fma_mixlo_f16, instead of fp32 -> fp16: <= 1 fp16 ulp on ties)
        assert torch.equal(x2[:B], x2[B:]) and ((x2[:B].float() - x).abs() <= x.abs() * 2 ** -10 + 1e-7).all()


def test_tiny_vae_decoder_vs_oracle_and_upsample_conv():
    from oracle.vae_encoder import VAEConfig
    from textboost_amd.vae import VAEGeometry
    geo = VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    B, h, w = 2, 8, 8
    ref, hip = _decoder_pair(VAEConfig.tiny(), geo, 12, B, h, w)
    z = torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(1)) * 0.18215
    with torch.no_grad():
        img_ref = ref.decode_latents(z)
    img = hip.decode(z.to(dev))
    assert img.shape == (B, 3, 64, 64) and img.min() >= 0 and img.max() <= 1
    assert (img.cpu() - img_ref).abs().max() < 2e-2 and rel_err(img, img_ref) < 5e-3, ((img.cpu() - img_ref).abs().max(), rel_err(img, img_ref))


def test_sd_vae_decoder_full_architecture_vs_oracle():
    """The real SD VAE decoder (49.49 M parameters, random init) at B=1, 16x16 latents -> 128x128 image."""
    from oracle.vae_encoder import VAEConfig
    from textboost_amd.vae import VAEGeometry
    B, h, w = 1, 16, 16
    ref, hip = _decoder_pair(VAEConfig.sd(), VAEGeometry(), 13, B, h, w)
    z = torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(2)) * 0.18215
    with torch.no_grad():
        pre = ref(z / 0.18215)
        img_ref = (pre / 2 + 0.5).clamp(0, 1)
    img = hip.decode(z.to(dev))
    assert rel_err(img, img_ref) < 1e-2, rel_err(img, img_ref)


def test_guided_sampling_loop_matches_oracle_on_tiny_models():
    """text hidden states -> 8 DPM-Solver++ steps of classifier-free guidance with the tiny UNet -> tiny VAE decoder, vs the oracle loop."""
    from oracle.sampler import sample_latents
    from oracle.vae_encoder import VAEConfig
    from tests.test_gpu_model import make_unet
    from textboost_amd.sampler import HipSampler
    from textboost_amd.vae import VAEGeometry
    B, hw, D, steps = 2, 16, 64, 8
    ref_unet, hip_unet, _ = make_unet(2 * B, hw, D, seed=5)
    geo = VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    ref_dec, hip_dec = _decoder_pair(VAEConfig.tiny(), geo, 14, B, hw, hw)
    g = torch.Generator().manual_seed(7)
    cond = torch.randn(B, 77, D, generator=g).half().float()
    uncond = torch.randn(B, 77, D, generator=g).half().float() * 0.5
    lat = torch.randn(B, 4, hw, hw, generator=g)
    with torch.no_grad():
        x_ref = sample_latents(lambda x, t, e: ref_unet(x.half().float(), t, e), cond, uncond, lat, steps=steps, guidance=7.5)
        img_ref = ref_dec.decode_latents(x_ref)
    smp = HipSampler(hip_unet, hip_dec, steps=steps, guidance=7.5)
    x = smp.denoise(cond.view(B * 77, D).to(dev), uncond.view(B * 77, D).to(dev), latents=lat.to(dev)).clone()
    e = rel_err(x, x_ref)
    assert e < 3e-2, f"latents after {steps} guided steps: rel-L2 {e}"   # fp16 UNet error amplified by guidance 7.5 over 8 steps
    img = smp.sample(cond.view(B * 77, D).to(dev), uncond.view(B * 77, D).to(dev), latents=lat.to(dev))
    assert img.shape == (B, 3, 8 * hw, 8 * hw) and torch.isfinite(img).all()
    assert (img.cpu() - img_ref).abs().mean() < 2e-2


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_ddpm_validation_scheduler_follows_the_oracle(pred):
    """--validation_scheduler DDPMScheduler (train_textboost.py:341-345, :483-495): the ancestral step in `tb_dpm_step`'s form (the variance noise
    rides in the m_prev operand) against the oracle's restatement of diffusers' DDPMScheduler.step over all 25 steps, then the guided loop
    with the tiny UNet against the oracle loop on the same noise draws."""
    from oracle.sampler import DDPMAncestral as Ref, sample_latents_ddpm
    from oracle.vae_encoder import VAEConfig
    from tests.test_gpu_model import make_unet
    from textboost_amd import ops
    from textboost_amd.sampler import DDPMAncestral, HipSampler
    from textboost_amd.vae import VAEGeometry
    ref = Ref(prediction_type=pred)
    sch = DDPMAncestral.from_config({"prediction_type": pred, "clip_sample": False, "steps_offset": 1})
    ts = ref.set_timesteps(25)
    assert sch.set_timesteps(25) == ts.tolist() and ts[0].item() == 961 and ts[-1].item() == 1
    torch.manual_seed(0)
    B, n, g = 2, 4 * 16 * 16, 7.5
    x_ref = torch.randn(B, n)
    x = x_ref.clone().to(dev); m_prev = torch.zeros(B, n, device=dev); x2 = torch.zeros(2 * B, n, device=dev, dtype=torch.float16)
    for i, t in enumerate(ts.tolist()):
        e2 = torch.randn(2 * B, n).half()
        z = torch.randn(B, n)
        eps = e2[:B].float() + g * (e2[B:].float() - e2[:B].float())
        x_ref = ref.step(eps, t, x_ref, z)
        a_t, s_t = sch.data_prediction_scalars(i)
        ca, cb, cc = sch.coefficients(i)
        m_prev.copy_(z)
        ops.dpm_step(x, e2.to(dev), m_prev, x2, n, B, g, a_t, s_t, ca, cb, cc)
    assert rel_err(x, x_ref) < 1e-5
    if pred != "epsilon":
        return
    Bm, hw, D, steps = 2, 16, 64, 6
    ref_unet, hip_unet, _ = make_unet(2 * Bm, hw, D, seed=5)
    _, hip_dec = _decoder_pair(VAEConfig.tiny(), VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1), 14, Bm, hw, hw)
    gen = torch.Generator().manual_seed(8)
    cond = torch.randn(Bm, 77, D, generator=gen).half().float()
    uncond = torch.randn(Bm, 77, D, generator=gen).half().float() * 0.5
    lat = torch.randn(Bm, 4, hw, hw, generator=gen)
    noise = [torch.randn(Bm, 4, hw, hw, generator=gen) for _ in range(steps)]
    with torch.no_grad():
        xr = sample_latents_ddpm(lambda x_, t_, e_: ref_unet(x_.half().float(), t_, e_), cond, uncond, lat, noise, steps=steps, guidance=7.5)
    smp = HipSampler(hip_unet, hip_dec, steps=steps, guidance=7.5, scheduler="DDPMScheduler")
    xs = smp.denoise(cond.view(Bm * 77, D).to(dev), uncond.view(Bm * 77, D).to(dev), latents=lat.to(dev), step_noise=[z.to(dev) for z in noise])
    assert rel_err(xs, xr) < 3e-2
    img = smp.sample(cond.view(Bm * 77, D).to(dev), uncond.view(Bm * 77, D).to(dev), latents=lat.to(dev))   # noise from the sampler's generator
    assert torch.isfinite(img).all()


def test_cli_validation_writes_image_grid(tmp_path):
    """--validation_prompts with tokenised prompts: 25-step guided sampling + VAE decode every --validation_steps, validation_{step}.jpg."""
    import os
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    from textboost_amd.workload import synthetic_ids
    data = tmp_path / "data"
    data.mkdir()
    g = torch.Generator().manual_seed(0)
    torch.save(synthetic_ids(2, [49408], g), str(data / "validation_input_ids.pt"))
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--instance_data_dir", str(data), "--output_dir", out,
                         "--train_batch_size", "2", "--resolution", "128", "--max_train_steps", "2", "--placeholder_token", "<dog>",
                         "--lora_rank", "4", "--mixed_precision", "fp16", "--seed", "3", "--validation_prompts", "a <dog>", "a <dog> on a beach",
                         "--validation_steps", "2", "--num_validation_images", "2"])
    T.main(args)
    img = Image.open(os.path.join(out, "validation_2.jpg"))
    assert img.size == (2 * 128, 2 * 128)
