"""Generates the augmentation fixtures, in the build container only.

Run:  python tests/golden/gen_augment_golden.py      (needs Pillow and /root/reference; never runs on the GPU box)

What is captured (data only -- inputs, call arguments and expected outputs, no reference source):
* augment_pil_primitives.npz   : Pillow's own results for `Image.resize` (BICUBIC / LANCZOS, up- and down-scaling, ragged sizes),
  `Image.transform(AFFINE)` (NEAREST / BICUBIC) and `ImageOps.grayscale(...).convert("RGB")` on seeded inputs.
* augment_reference_calls.json : seeded runs of the REAL `textboost/augment/paired_augmentation.py` (`PairedAugmentation.__call__`,
  imported from /root/reference by file path): for each case the arguments the reference passed to `v2.functional.pad / affine /
  center_crop` (torchvision is absent, so a RECORDING stand-in logs them and forwards to Pillow following torchvision's published
  PIL code path), the returned prompt, the output size and a CRC of the output pixels.
* augment_reference_images.npz : the input and output pixels of those cases.
Cases whose op chain never enters the stand-in (crop, grayscale, horizontal_flip, square_photo_collage) are the reference + Pillow
only; "glue": true marks the ones that went through the restated torchvision glue.
"""
import importlib.util
import json
import math
import os
import random
import sys
import types
import zlib

import numpy as np
from PIL import Image, ImageOps

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
CALLS = []


# ----------------------------------------------------------------- recording stand-in for torchvision.transforms.v2.functional
def _inv_affine(center, angle, translate, scale, shear):
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [x / scale for x in [d, -b, 0.0, -c, a, 0.0]]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _pad(image, padding, fill=None, padding_mode="constant"):
    CALLS.append(["pad", [int(p) for p in padding], padding_mode])
    lr, tb = padding
    a = np.asarray(image)
    mode = {"edge": "edge", "constant": "constant"}[padding_mode]
    return Image.fromarray(np.pad(a, ((tb, tb), (lr, lr), (0, 0)), mode=mode))


def _affine(image, angle, translate, scale, shear, interpolation=0, fill=None, center=None):
    CALLS.append(["affine", float(angle), [float(t) for t in translate], float(scale), float(shear), int(interpolation)])
    w, h = image.size
    m = _inv_affine([w * 0.5, h * 0.5], float(angle), [float(t) for t in translate], scale, [float(shear), 0.0])
    return image.transform((w, h), Image.AFFINE, m, int(interpolation))


def _center_crop(image, output_size):
    CALLS.append(["center_crop", [int(s) for s in output_size]])
    ch, cw = output_size
    w, h = image.size
    if ch > h or cw > w:
        pl = (cw - w) // 2 if cw > w else 0
        pt = (ch - h) // 2 if ch > h else 0
        pr = (cw - w + 1) // 2 if cw > w else 0
        pb = (ch - h + 1) // 2 if ch > h else 0
        image = Image.fromarray(np.pad(np.asarray(image), ((pt, pb), (pl, pr), (0, 0))))
        w, h = image.size
        if cw == w and ch == h:
            return image
    top = int(round((h - ch) / 2.0))
    left = int(round((w - cw) / 2.0))
    return image.crop((left, top, left + cw, top + ch))


def load_reference_module():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    v2 = types.ModuleType("torchvision.transforms.v2")
    v2.functional = types.SimpleNamespace(pad=_pad, affine=_affine, center_crop=_center_crop)
    tv.transforms, tr.v2 = tr, v2
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.v2": v2})
    spec = importlib.util.spec_from_file_location("ref_paired_augmentation", f"{REF}/textboost/augment/paired_augmentation.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_image(seed, h, w):
    """Seeded image with structure (gradients + blocks) and noise, so resampling is exercised on edges and on flat areas."""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx // 7 + yy // 5) % 2) * 200], -1)
    noise = r.integers(-40, 41, (h, w, 3))
    return np.clip(base + noise, 0, 255).astype(np.uint8)


def make_primitives():
    out = {}
    cases = []
    for i, (h, w, oh, ow) in enumerate([(37, 53, 16, 16), (64, 64, 64, 32), (100, 80, 33, 47), (30, 30, 77, 91), (257, 129, 64, 64),
                                        (20, 20, 20, 20), (9, 7, 3, 2), (5, 5, 40, 40), (200, 300, 64, 96), (1, 1, 4, 4), (160, 120, 64, 85)]):
        a = np.random.default_rng(100 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"resize_in_{i}"] = a
        for name, f in (("bicubic", Image.BICUBIC), ("lanczos", Image.LANCZOS)):
            out[f"resize_{name}_{i}"] = np.asarray(Image.fromarray(a).resize((ow, oh), f))
        cases.append([h, w, oh, ow])
    out["resize_cases"] = np.array(cases)
    acases = []
    for i, (h, w, s, tx) in enumerate([(40, 40, 0.34, 0), (33, 50, 0.5, 0), (64, 48, 0.77, 0), (40, 40, 1.23, 0), (33, 50, 1.4, 0),
                                       (40, 40, 1.0, -7), (33, 50, 1.0, 13), (48, 64, 0.9, 3.5)]):
        a = np.random.default_rng(200 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        m = _inv_affine([w * 0.5, h * 0.5], 0.0, [float(tx), 0.0], s, [0.0, 0.0])
        out[f"affine_in_{i}"] = a
        out[f"affine_matrix_{i}"] = np.array(m, np.float64)
        out[f"affine_nearest_{i}"] = np.asarray(Image.fromarray(a).transform((w, h), Image.AFFINE, m, Image.NEAREST))
        out[f"affine_bicubic_{i}"] = np.asarray(Image.fromarray(a).transform((w, h), Image.AFFINE, m, Image.BICUBIC))
        acases.append([h, w])
    out["affine_cases"] = np.array(acases)
    a = np.random.default_rng(300).integers(0, 256, (50, 60, 3), dtype=np.uint8)
    out["gray_in"] = a
    out["gray_out"] = np.asarray(ImageOps.grayscale(Image.fromarray(a)).convert("RGB"))
    np.savez_compressed(os.path.join(OUT, "augment_pil_primitives.npz"), **out)
    print("primitives:", len(out), "arrays")


def make_reference_calls():
    mod = load_reference_module()
    records, images = [], {}
    shapes = [(48, 40), (40, 56), (64, 64), (36, 36)]
    n = 0
    for cfg_i, cfg in enumerate([dict(hflip="inversion", inversion=True, p=0.9, color_prob=0.5, augment_prompt=True, ops="object"),
                                 dict(hflip="false", inversion=False, p=0.9, color_prob=0.5, augment_prompt=True, ops="object"),
                                 dict(hflip="true", inversion=False, p=0.5, color_prob=0.2, augment_prompt=False, ops="object"),
                                 dict(hflip="inversion", inversion=True, p=0.5, color_prob=0.2, augment_prompt=True, ops="style")]):
        pipe = mod.PairedAugmentation(**cfg)
        for seed in range(14 if cfg_i < 2 else 6):
            h, w = shapes[seed % len(shapes)]
            img = test_image(1000 + n, h, w)
            np.random.seed(seed * 7 + cfg_i)
            random.seed(seed * 11 + cfg_i)
            CALLS.clear()
            out, prompt, mask = pipe(Image.fromarray(img), "a photo of <sks> dog")
            assert mask is None
            o = np.asarray(out.convert("RGB"))
            records.append({"case": n, "config": cfg, "np_seed": seed * 7 + cfg_i, "py_seed": seed * 11 + cfg_i, "in_hw": [h, w],
                            "image_seed": 1000 + n, "calls": json.loads(json.dumps(CALLS)), "glue": len(CALLS) > 0,
                            "prompt": str(prompt), "out_hw": list(o.shape[:2]), "crc32": zlib.crc32(o.tobytes())})
            images[f"in_{n}"] = img
            images[f"out_{n}"] = o
            n += 1
    with open(os.path.join(OUT, "augment_reference_calls.json"), "w") as f:
        json.dump({"prompt_in": "a photo of <sks> dog", "records": records}, f, indent=0)
    np.savez_compressed(os.path.join(OUT, "augment_reference_images.npz"), **images)
    ops = {}
    for r in records:
        k = "+".join(c[0] for c in r["calls"]) or "no-glue"
        ops[k] = ops.get(k, 0) + 1
    print("reference cases:", n, ops)


if __name__ == "__main__":
    make_primitives()
    make_reference_calls()
