"""VAE encoder (train_textboost.py:1036-1037) on the GPU, through the C-ABI, against the fp32 CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _pair(cfg_oracle, geo, seed, B, H, W):
    from oracle.vae_encoder import VAEEncoder
    from textboost_amd import models
    from textboost_amd.vae import HipVAEEncoder, vae_encoder_shapes
    sd = models.random_state_dict(vae_encoder_shapes(geo), seed, device="cpu")
    sd = {k: v.half().float() for k, v in sd.items()}  # both sides see fp16-representable weights
    with torch.device("meta"):
        ref = VAEEncoder(cfg_oracle)
    ref = ref.to_empty(device="cpu")
    ref.load_state_dict(sd)
    hip = HipVAEEncoder(geo, {k: v.to(dev) for k, v in sd.items()}, B, H, W, device=dev)
    return ref, hip


def test_asymmetric_stride2_conv_and_rgb_conv_in_and_softmax():
    from textboost_amd import ops
    from textboost_amd.unet import pack_conv3x3
    torch.manual_seed(0)
    B, C, Co, H, W = 2, 64, 128, 12, 16
    x = torch.randn(B, C, H, W, device=dev).half()
    w = (torch.randn(Co, C, 3, 3, device=dev) / 24).half()
    bias = torch.randn(Co, device=dev)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), bias, stride=2)        # diffusers Downsample2D(padding=0)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()
    out = torch.empty(B * (H // 2) * (W // 2), Co, device=dev, dtype=torch.float16)
    geo = dict(B=B, Hin=H, Win=W, Cin=C, Hout=H // 2, Wout=W // 2, stride=2, sign=1, upsample=0, transposed=0, shift=1)
    ops.gemm(xn, pack_conv3x3(w, dev)[0], out, conv=geo, bias=bias)
    got = out.view(B, H // 2, W // 2, Co).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 2e-3
    # RGB conv_in: NCHW fp32 [B,3,H,W] -> NHWC fp16
    px = torch.rand(B, 3, H, W, device=dev) * 2 - 1
    w3 = torch.randn(Co, 3, 3, 3, device=dev) / 5
    ref = F.conv2d(px, w3, bias, padding=1)
    out = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    ops.convin_to_nhwc(px, 3, w3.permute(2, 3, 1, 0).reshape(27, Co).contiguous(), bias, out, B, H, W, Co)
    assert rel_err(out.view(B, H, W, Co).permute(0, 3, 1, 2), ref) < 1e-3
    # row softmax, fp32 scores -> fp16 probabilities
    s = torch.randn(70, 4096, device=dev) * 6
    p = torch.empty(70, 4096, device=dev, dtype=torch.float16)
    ops.softmax_rows(s, p)
    assert rel_err(p, torch.softmax(s, -1)) < 1e-3 and (p.float().sum(-1) - 1).abs().max() < 2e-3


def test_tiny_vae_encoder_vs_oracle():
    from oracle.vae_encoder import VAEConfig
    from textboost_amd.vae import VAEGeometry
    geo = VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    B, H, W = 2, 64, 64
    ref, hip = _pair(VAEConfig.tiny(), geo, 5, B, H, W)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    eps = torch.randn(B, 4, H // 8, W // 8, generator=g)
    with torch.no_grad():
        mean, logvar = ref.moments(x)
        z_ref = ref.encode_sample(x, noise=eps)
    mom, h, w = hip.moments(x.to(dev))
    mom = mom.view(B, h, w, 8).permute(0, 3, 1, 2)
    assert rel_err(mom[:, :4], mean) < 5e-3, rel_err(mom[:, :4], mean)
    assert (mom[:, 4:].cpu() - logvar).abs().max() < 5e-2  # logvar enters through exp(0.5 x): absolute tolerance
    z = hip.encode(x.to(dev), noise=eps.to(dev))
    assert z.shape == (B, 4, H // 8, W // 8) and z.dtype == torch.float32
    assert rel_err(z, z_ref) < 5e-3, rel_err(z, z_ref)
    # zero noise -> scaled mean; drawing inside: finite, right variance class, reproducible under a seeded generator
    assert rel_err(hip.encode(x.to(dev), noise=torch.zeros_like(eps).to(dev)), mean * 0.18215) < 5e-3
    hip.generator = torch.Generator(device=dev).manual_seed(11)
    z1 = hip.encode(x.to(dev)).clone()
    hip.generator = torch.Generator(device=dev).manual_seed(11)
    z2 = hip.encode(x.to(dev)).clone()
    assert torch.equal(z1, z2) and torch.isfinite(z1).all()


def test_sd_vae_encoder_full_architecture_vs_oracle():
    """The real SD VAE encoder (34.16 M parameters, random init) at B=1, 256x256 (the oracle needs ~0.28 TFLOP of fp32 on the host)."""
    from oracle.vae_encoder import VAEConfig
    from textboost_amd.vae import VAEGeometry
    B, H, W = 1, 256, 256
    ref, hip = _pair(VAEConfig.sd(), VAEGeometry(), 9, B, H, W)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    eps = torch.randn(B, 4, H // 8, W // 8, generator=g)
    with torch.no_grad():
        z_ref = ref.encode_sample(x, noise=eps)
    z = hip.encode(x.to(dev), noise=eps.to(dev))
    e = rel_err(z, z_ref)
    assert e < 1e-2, f"SD VAE encoder latents rel-L2 {e}"


def test_sd_vae_encoder_metric_shapes_properties():
    """B=8, 512x512 (the metric's image size): finite, the right shape, batch rows independent, bit-reproducible."""
    from textboost_amd import models
    from textboost_amd.vae import HipVAEEncoder, VAEGeometry, vae_encoder_shapes
    geo = VAEGeometry()
    sd = models.random_state_dict(vae_encoder_shapes(geo), 21, device=dev)
    B = 8
    enc = HipVAEEncoder(geo, sd, B, 512, 512, device=dev)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand(B, 3, 512, 512, device=dev, generator=g) * 2 - 1
    eps = torch.randn(B, 4, 64, 64, device=dev, generator=g)
    z = enc.encode(x, noise=eps).clone()
    assert z.shape == (B, 4, 64, 64) and torch.isfinite(z).all()
    assert torch.equal(z, enc.encode(x, noise=eps))
    xp = x.clone(); xp[3] = x[5]                      # rows of the batch do not mix
    zp = enc.encode(xp, noise=eps)
    assert torch.equal(zp[0], z[0]) and not torch.equal(zp[3], z[3])
    one = HipVAEEncoder(geo, sd, 1, 512, 512, device=dev)
    z5 = one.encode(x[5:6].contiguous(), noise=eps[5:6].contiguous())
    assert rel_err(z5, z[5:6]) < 2e-3                  # same image alone (different GroupNorm chunking / tile schedule)


def test_training_step_from_pixels_matches_oracle_and_graph_replay():
    """train_textboost.py:1027-1090 with the VAE attached: pixels -> latents -> noisy -> UNet -> loss, eager vs the oracle chain,
    then HIP-graph replay == eager bit for bit."""
    from oracle import train_step as ts
    from oracle.vae_encoder import VAEConfig
    from tests.test_gpu_model import build_step
    from textboost_amd.vae import VAEGeometry
    B, hw, D = 2, 16, 64
    geo = VAEGeometry(block_out_channels=(64, 64, 128, 128), layers_per_block=1)
    g = torch.Generator().manual_seed(8)
    px = torch.rand(B, 3, 8 * hw, 8 * hw, generator=g) * 2 - 1
    eps = torch.randn(B, 4, hw, hw, generator=g)
    noise = torch.randn(B, 4, hw, hw, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    results = []
    for mode in ("eager", "graph"):
        st_ref, step, added = build_step(B, hw, D)
        vref, vhip = _pair(VAEConfig.tiny(), geo, 6, B, 8 * hw, 8 * hw)
        step.attach_vae(vhip)
        gi = torch.Generator().manual_seed(9)
        ids, pids = ts.synthetic_ids(B, added, gi), ts.synthetic_ids(B, added, gi, prior=True)
        step.pixel_values.copy_(px); step.vae_eps.copy_(eps); step.noise.copy_(noise); step.timesteps.copy_(t)
        step.input_ids.copy_(ids); step.prior_ids.copy_(pids)
        if mode == "eager":
            with torch.no_grad():
                x0_ref = vref.encode_sample(px, noise=eps)
            out = st_ref.step(x0_ref, noise, t, ids, pids)
            step.step_eager()
            assert rel_err(step.x0, x0_ref) < 5e-3
            sc = step.scalars()
            assert abs(sc["loss_mse"] - out["mse"]) < 2e-2 * abs(out["mse"]) + 1e-4, (sc, out["mse"])
            step.step_eager()
        else:
            step.capture(warmup=1)
            step.replay()
        torch.cuda.synchronize()
        results.append((step.x0.clone(), step.te.lora_A.clone(), step.te.lora_B.clone(), step.te.token_table[49408:].clone(), step.state.clone()))
    for a, b in zip(*results):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_cli_trains_from_pixel_values(tmp_path):
    """train_textboost.py fed `pixel_values.pt` (what the reference's dataset yields): the VAE encoder runs on the device each step."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    data = tmp_path / "data"
    data.mkdir()
    g = torch.Generator().manual_seed(0)
    torch.save(torch.rand(3, 3, 128, 128, generator=g) * 2 - 1, str(data / "pixel_values.pt"))
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--instance_data_dir", str(data), "--output_dir", out,
                         "--train_batch_size", "2", "--resolution", "128", "--max_train_steps", "3", "--placeholder_token", "<dog>",
                         "--lora_rank", "4", "--mixed_precision", "fp16", "--seed", "1"])
    T.main(args)
    log = open(os.path.join(out, "training.log")).read()
    assert "VAE" in log and os.path.exists(os.path.join(out, "dog.bin"))
    d = torch.load(os.path.join(out, "dog.bin"))
    assert torch.isfinite(d["<dog>"]).all()
