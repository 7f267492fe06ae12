import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textboost_amd import models
from textboost_amd.sampler import HipSampler
from textboost_amd.unet import HipUNet
from textboost_amd.vae import HipVAEDecoder, VAEGeometry, vae_decoder_shapes
dev = "cuda"; n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
usd = models.random_state_dict(models.unet_shapes(models.SD15_UNET), 1, device=dev)
unet = HipUNet(models.SD15_UNET, usd, 2 * n, 64, 64, device=dev); del usd
dec = HipVAEDecoder(VAEGeometry(), models.random_state_dict(vae_decoder_shapes(VAEGeometry()), 2, device=dev), n, 64, 64, device=dev)
smp = HipSampler(unet, dec, steps=25, guidance=7.5)
cond = torch.randn(n * 77, 768, device=dev).half(); uncond = torch.randn(n * 77, 768, device=dev).half()
img = smp.sample(cond, uncond); torch.cuda.synchronize()
t0 = time.perf_counter(); img = smp.sample(cond, uncond); torch.cuda.synchronize(); t1 = time.perf_counter()
lat = smp.denoise(cond, uncond); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{n} images 512^2, 25 DPM-Solver++ steps, guidance 7.5 (eager launches): {t1-t0:.3f} s total ({n/(t1-t0):.2f} images/s); denoise only {t2-t1:.3f} s; decode {(t1-t0)-(t2-t1):.3f} s; finite {torch.isfinite(img).all().item()}")
