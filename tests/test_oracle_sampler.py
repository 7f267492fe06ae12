"""CPU tests of the sampling-path oracle: published VAE decoder size, DPM-Solver++(2M) identities."""
import torch

from oracle.sampler import DPMSolverPP2M, VAEDecoder, count_decoder_params, sample_latents
from oracle.vae_encoder import VAEConfig, count_encoder_params


def test_published_vae_parameter_counts():
    dec, pq = count_decoder_params(VAEConfig.sd())
    enc, q = count_encoder_params(VAEConfig.sd())
    assert dec == 49_490_179 and pq == 20
    assert enc + q + dec + pq == 83_653_863  # the SD1.x / SD2.x AutoencoderKL


def test_decoder_shapes_and_postprocess():
    torch.manual_seed(0)
    m = VAEDecoder(VAEConfig.tiny())
    z = torch.randn(2, 4, 8, 8) * 0.18215
    img = m.decode_latents(z)
    assert img.shape == (2, 3, 64, 64) and img.min() >= 0 and img.max() <= 1
    torch.testing.assert_close(img, (m(z / 0.18215) / 2 + 0.5).clamp(0, 1))


def test_dpm_solver_timestep_spacings():
    """DPMSolverMultistepScheduler.set_timesteps [3P, diffusers 0.29]: `from_config` on SD's PNDM / DDIM scheduler instance inherits
    timestep_spacing "leading" + steps_offset 1 -> 951, 913, ..., 39 for 25 steps (step_ratio = 1000 // 26)."""
    ts = DPMSolverPP2M().set_timesteps(25)
    assert ts.tolist() == [951 - 38 * i for i in range(25)]
    ts = DPMSolverPP2M(timestep_spacing="trailing").set_timesteps(25)
    assert ts.tolist() == [999 - 40 * i for i in range(25)]
    ts = DPMSolverPP2M(timestep_spacing="leading", steps_offset=0).set_timesteps(10)
    assert ts.tolist() == [900 - 90 * i for i in range(10)]


def test_dpm_solver_v_prediction_point_mass_exactness():
    """SD2.1-768: the model predicts v = alpha_t eps - sigma_t x0; with the exact v of a point mass the solver must stay on x_t."""
    sch = DPMSolverPP2M(prediction_type="v_prediction")
    sch.set_timesteps(20)
    c = torch.tensor([0.7, -1.3, 2.0]); e = torch.tensor([0.3, 0.1, -0.9])
    a0, s0 = sch._alpha_sigma(sch.sigmas[0])
    x = (a0 * c + s0 * e).float()
    for i in range(20):
        a_t, s_t = sch._alpha_sigma(sch.sigmas[i])
        eps = (x - float(a_t) * c) / float(s_t)
        v = float(a_t) * eps - float(s_t) * c
        x = sch.step(v, x)
        a_n, s_n = sch._alpha_sigma(sch.sigmas[i + 1])
        torch.testing.assert_close(x, (a_n * c + s_n * e).float(), rtol=1e-4, atol=1e-4)


def test_dpm_solver_timesteps_and_point_mass_exactness():
    sch = DPMSolverPP2M(timestep_spacing="linspace")
    ts = sch.set_timesteps(25)
    assert ts[0].item() == 999 and len(ts) == 25 and (ts[:-1] > ts[1:]).all() and ts[-1].item() == 40
    assert sch.sigmas[-1] == 0 and abs(sch.sigmas[0].item() - 14.6146) < 1e-3        # sqrt((1 - abar_999) / abar_999)
    # a model that returns the exact noise of a point mass at c: every consistent solver reproduces x_t = alpha_t c + sigma_t eps
    c = torch.tensor([0.7, -1.3, 2.0])
    e = torch.tensor([0.3, 0.1, -0.9])
    a0, s0 = sch._alpha_sigma(sch.sigmas[0])
    x = (a0 * c + s0 * e).float()
    for i in range(25):
        a_t, s_t = sch._alpha_sigma(sch.sigmas[i])
        eps = (x - float(a_t) * c) / float(s_t)
        x = sch.step(eps, x)
        a_n, s_n = sch._alpha_sigma(sch.sigmas[i + 1])
        torch.testing.assert_close(x, (a_n * c + s_n * e).float(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x, c, rtol=1e-4, atol=1e-4)
    # coefficient identities: first-order steps have no history term; the second-order weights of m0 and m_prev sum to the first-order one
    sch.set_timesteps(25)
    assert sch.coefficients(0)[2] == 0.0 and sch.coefficients(24) == (0.0, 1.0, 0.0)
    a, b, cc = sch.coefficients(7)
    s0_, s1_ = sch.sigmas[7], sch.sigmas[8]
    al1, st1 = sch._alpha_sigma(s1_); al0, st0 = sch._alpha_sigma(s0_)
    h = (torch.log(al1) - torch.log(st1)) - (torch.log(al0) - torch.log(st0))
    assert abs((b + cc) - float(-al1 * (torch.exp(-h) - 1))) < 1e-9 and abs(a - float(st1 / st0)) < 1e-12


def test_guidance_loop_reduces_to_conditional_model_at_scale_one():
    torch.manual_seed(1)
    W = torch.randn(4, 4) * 0.1
    def unet(x, t, ehs):  # a linear "UNet" whose output depends on the conditioning
        return torch.einsum("oc,bchw->bohw", W, x) + ehs.mean(dim=(1, 2)).view(-1, 1, 1, 1)
    lat = torch.randn(2, 4, 8, 8)
    cond, uncond = torch.randn(2, 77, 16), torch.zeros(2, 77, 16)
    x1 = sample_latents(unet, cond, uncond, lat, steps=6, guidance=1.0)
    x1c = sample_latents(lambda x, t, e: unet(x, t, torch.cat([cond, cond])), cond, cond, lat, steps=6, guidance=1.0)
    torch.testing.assert_close(x1, x1c)
    assert torch.isfinite(sample_latents(unet, cond, uncond, lat, steps=6, guidance=7.5)).all()
