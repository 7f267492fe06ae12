"""Feeder / augmentation measurement: per-kernel roofline (HBM-bound byte work), whole-batch feeder rate at the metric's
shape (batch 8, 512^2, camera-sized instance images), and the CPU side (Pillow, what the reference runs per sample on DataLoader workers).
Usage: python scratch/bench_augment.py [--out gpurun_out/bench_augment.json]"""
import argparse, json, os, random, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textboost_amd import augment as D, ops

HBM_PEAK = 8000.0  # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


class WordTokenizer:
    model_max_length = 77
    def __call__(self, prompt, truncation=True, padding="max_length", max_length=77, return_tensors="pt"):
        ids = ([49406] + [sum(map(ord, w)) % 49405 for w in prompt.split()])[:max_length - 1]
        return types.SimpleNamespace(input_ids=torch.tensor([ids + [49407] * (max_length - len(ids))], dtype=torch.int64))


def image(seed, h, w):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    return np.clip(np.stack([xx * 255 // w, yy * 255 // h, (xx + yy) % 256], -1) + r.integers(-30, 31, (h, w, 3)), 0, 255).astype(np.uint8)


def ev_time(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--h", type=int, default=1536)
    ap.add_argument("--w", type=int, default=2048)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    H, W, R, B = args.h, args.w, args.size, args.batch
    host = [image(i, H, W) for i in range(2)]
    dev = [D.to_device_image(a) for a in host]
    t = dev[0]
    res = {"image": [H, W], "size": R, "batch": B, "kernels": {}}
    # ---- per-kernel: Lanczos short-edge resize = horizontal pass (W -> nw) then vertical pass (H -> R)
    nw, nh = (R, int(R * H / W)) if W <= H else (int(R * W / H), R)
    _, bh, kh = ops.resample_coeffs(W, nw, D.LANCZOS)
    _, bv, kv = ops.resample_coeffs(H, nh, D.LANCZOS)
    bh, kh, bv, kv = bh.cuda(), kh.cuda(), bv.cuda(), kv.cuda()
    mid = ops.img_resample(t, nw, bh, kh, 0, True)
    dt = ev_time(lambda: ops.img_resample(t, nw, bh, kh, 0, True))
    alg = (H * W + H * nw) * 4
    res["kernels"]["resample_kernel<0> (horizontal, Lanczos %d->%d, %d rows, ksize %d)" % (W, nw, H, kh.shape[1])] = {
        "us": dt * 1e6, "alg_bytes": alg, "GBps": alg / dt / 1e9, "frac_hbm": alg / dt / 1e9 / HBM_PEAK}
    dt = ev_time(lambda: ops.img_resample(mid, nh, bv, kv, 1, True))
    alg = (H * nw + nh * nw) * 4
    res["kernels"]["resample_kernel<1> (vertical, Lanczos %d->%d, %d cols, ksize %d)" % (H, nh, nw, kv.shape[1])] = {
        "us": dt * 1e6, "alg_bytes": alg, "GBps": alg / dt / 1e9, "frac_hbm": alg / dt / 1e9 / HBM_PEAK}
    m = D.inverse_affine_matrix([W * 0.5, H * 0.5], 0.0, [0.0, 0.0], 1.17, [0.0, 0.0])
    dt = ev_time(lambda: ops.img_affine_bicubic(t, m, 0, 0, 0, 0, W, H))
    alg = 2 * H * W * 4
    res["kernels"]["affine_bicubic_kernel (scale 1.17, %dx%d, fp64)" % (W, H)] = {"us": dt * 1e6, "alg_bytes": alg, "GBps": alg / dt / 1e9,
                                                                                  "frac_hbm": alg / dt / 1e9 / HBM_PEAK}
    xt, yt = torch.arange(W - 1, -1, -1, dtype=torch.int32).cuda(), torch.arange(H, dtype=torch.int32).cuda()
    dt = ev_time(lambda: ops.img_gather(t, xt, yt, True))
    res["kernels"]["gather_kernel (flip + luma, %dx%d)" % (W, H)] = {"us": dt * 1e6, "alg_bytes": alg, "GBps": alg / dt / 1e9,
                                                                     "frac_hbm": alg / dt / 1e9 / HBM_PEAK}
    small = D.resize_short_edge(t, R)
    dst = torch.empty(3, R, R, device="cuda")
    dt = ev_time(lambda: ops.img_to_pixels(small, 0, 0, dst))
    alg = R * R * 4 + 3 * R * R * 4
    res["kernels"]["to_pixels_kernel (%d^2)" % R] = {"us": dt * 1e6, "alg_bytes": alg, "GBps": alg / dt / 1e9, "frac_hbm": alg / dt / 1e9 / HBM_PEAK}
    # ---- whole feeder batch (host draws + tables + launches), the reference driver's augmentation settings
    templates = ["{}", "a {}", "one {}", "the {}", "photo of a {}"]
    for name, pipe in (("no augmentation", None), ("paug p=0.8 inversion", D.PairedAugmentation(hflip="inversion", inversion=True, p=0.8))):
        feeder = D.DeviceFeeder([(d, ["<sks>"]) for d in dev], WordTokenizer(), templates, size=R, center_crop=False, augment_pipe=pipe)
        out = torch.empty(B, 3, R, R, device="cuda")
        random.seed(0), np.random.seed(0), torch.manual_seed(0)
        for _ in range(5):
            feeder.batch(list(range(B)), out=out)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            feeder.batch(list(range(B)), out=out)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        res["feeder batch, " + name] = {"ms_per_batch_host_enqueue": t_host * 1e3, "ms_per_batch_incl_gpu": t_all * 1e3,
                                        "samples_per_s": B / t_all}
    # ---- CPU side: Pillow itself (the reference's per-sample work), one core
    try:
        from PIL import Image
        import PIL
        im = Image.fromarray(host[0])
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 5.0:
            r = im.resize((nw, nh), Image.LANCZOS)
            k += 1
        t_res = (time.perf_counter() - t0) / k
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 5.0:
            r = im.transform((W, H), Image.AFFINE, m, Image.BICUBIC)
            k += 1
        t_aff = (time.perf_counter() - t0) / k
        res["cpu_pillow"] = {"version": PIL.__version__, "cores": 1, "lanczos_resize_ms": t_res * 1e3, "affine_bicubic_ms": t_aff * 1e3,
                             "resizes_per_s": 1.0 / t_res, "note": "what one DataLoader worker of the reference spends per sample (resize only)"}
    except ImportError:
        res["cpu_pillow"] = None
    print(json.dumps(res, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
