"""CPU tests of the VAE-encoder oracle: published parameter counts and per-block equality with torch's own ops."""
import torch
import torch.nn.functional as F

from oracle.vae_encoder import AttnBlock, Downsample2D, ResnetBlock2D, VAEConfig, VAEEncoder, count_encoder_params


def test_published_parameter_counts():
    enc, quant = count_encoder_params(VAEConfig.sd())
    assert enc == 34_163_592 and quant == 72          # encoder + decoder 49,490,179 + quant 72 + post_quant 20 = 83,653,863
    from textboost_amd.vae import VAEGeometry, vae_encoder_shapes
    from textboost_amd.models import count_params
    assert count_params(vae_encoder_shapes(VAEGeometry())) == enc + quant
    # same keys / shapes as the oracle module (and therefore as diffusers' AutoencoderKL encoder half)
    with torch.device("meta"):
        sd = VAEEncoder(VAEConfig.sd()).state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == vae_encoder_shapes(VAEGeometry())


def test_blocks_match_torch_ops():
    torch.manual_seed(0)
    x = torch.randn(2, 64, 10, 12)
    d = Downsample2D(64)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), d.conv.weight, d.conv.bias, stride=2)
    assert d(x).shape == (2, 64, 5, 6) and torch.equal(d(x), ref)
    # same thing as an explicit gather: out(y, x) reads rows 2y + ky, cols 2x + kx (ky, kx in 0..2), zero beyond the border
    xp = torch.zeros(2, 64, 12, 14); xp[:, :, :10, :12] = x
    man = sum(torch.einsum("oc,bcyx->boyx", d.conv.weight[:, :, ky, kx], xp[:, :, ky:ky + 10:2, kx:kx + 12:2])
              for ky in range(3) for kx in range(3)) + d.conv.bias.view(1, -1, 1, 1)
    torch.testing.assert_close(d(x), man, rtol=1e-4, atol=1e-4)
    r = ResnetBlock2D(64, 128, 32, 1e-6)
    h = r.conv1(F.silu(F.group_norm(x, 32, r.norm1.weight, r.norm1.bias, 1e-6)))
    h = r.conv2(F.silu(F.group_norm(h, 32, r.norm2.weight, r.norm2.bias, 1e-6)))
    torch.testing.assert_close(r(x), r.conv_shortcut(x) + h)
    a = AttnBlock(64, 32, 1e-6)
    hn = F.group_norm(x, 32, a.group_norm.weight, a.group_norm.bias, 1e-6).flatten(2).transpose(1, 2)
    o = F.scaled_dot_product_attention(a.to_q(hn)[:, None], a.to_k(hn)[:, None], a.to_v(hn)[:, None])[:, 0]
    torch.testing.assert_close(a(x), x + a.to_out[0](o).transpose(1, 2).reshape(x.shape), rtol=1e-4, atol=1e-5)


def test_sample_is_reparameterised_gaussian():
    torch.manual_seed(1)
    m = VAEEncoder(VAEConfig.tiny())
    x = torch.rand(2, 3, 64, 64) * 2 - 1
    mean, logvar = m.moments(x)
    assert mean.shape == (2, 4, 8, 8) and logvar.min() >= -30 and logvar.max() <= 20
    eps = torch.randn(2, 4, 8, 8)
    z = m.encode_sample(x, noise=eps)
    torch.testing.assert_close(z, (mean + torch.exp(0.5 * logvar) * eps) * 0.18215)
    torch.testing.assert_close(m.encode_sample(x, noise=torch.zeros_like(eps)), mean * 0.18215)
