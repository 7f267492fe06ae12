"""ORACLE (test infrastructure only) -- numpy restatement of the reference's image-augmentation + feeder path, SURVEY.md 8(f) row 3:

    textboost/augment/paired_augmentation.py:12-351   PairedAugmentation and its ops (adjust_scale, crop, horizontal_translate,
                                                      horizontal_flip, grayscale, square_photo_collage)
    textboost/dataset.py:324-418                      Resize(size, LANCZOS) -> Center/RandomCrop -> ToImage/ToDtype/Normalize, the prompt
                                                      template draw, tokenize_prompt

The byte arithmetic of those ops lives in two third-party packages:
  * Pillow (the image's own install, 12.x; the reference pins none): `Image.resize` (libImaging/Resample.c: separable two-pass convolution,
    horizontal then vertical, 8-bit intermediate, coefficients in 22-bit fixed point), `Image.transform(AFFINE)` (libImaging/Geometry.c:
    `ImagingScaleAffine` table walk for NEAREST, `ImagingGenericTransform` + `bicubic_filter32RGB` in double precision, result truncated, for BICUBIC),
    `convert("L")` (libImaging/Convert.c: (19595 R + 38470 G + 7471 B + 0x8000) >> 16).  Pillow IS installed in this container, so every
    primitive below is pinned bit-exactly against it: tests/golden/augment_*.npz were produced by tests/golden/gen_augment_golden.py
    calling Pillow, and tests/test_oracle_augment.py re-checks live whenever PIL imports.
  * torchvision (`v2.functional.pad / affine / center_crop`, `v2.Resize`, `v2.RandomCrop.get_params`, `ToDtype(scale=True)`, `Normalize`):
    NOT installed, not vendored -- restated from its published code; that glue is "parity unpinned" (the Pillow calls it ends in are pinned).
The random-draw order and the prompt edits of `PairedAugmentation.__call__` are pinned against the reference itself: the fixture
tests/golden/augment_reference_calls.json records, for seeded runs of the real `paired_augmentation.py` (imported here with a
recording stand-in for the absent `torchvision.transforms.v2.functional`), the arguments it passes and the prompts it returns.

Images are uint8 arrays [H, W, 3].  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import math
import random as pyrandom
from typing import List, Optional, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c
NEAREST, LANCZOS, BILINEAR, BICUBIC = 0, 1, 2, 3  # PIL.Image.Resampling ids


# ------------------------------------------------------------------------------------------------ Image.resize (Resample.c)
def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos_filter(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


_FILTERS = {BICUBIC: (_bicubic_filter, 2.0), LANCZOS: (_lanczos_filter, 3.0)}


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int, filt: int):
    """Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc`: (ksize, bounds[out, 2] = (xmin, count), kk[out, ksize] int32)."""
    fn, fsupport = _FILTERS[filt]
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [fn((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, filt: int, axis: int) -> np.ndarray:
    """One pass of `ImagingResampleHorizontal_8bpc` / `Vertical_8bpc`: ss = 1 << 21; ss += px * k; out = clip8(ss >> 22)."""
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    n = a.shape[0]
    ksize, bounds, kk = precompute_coeffs(n, 0.0, float(n), out_size, filt)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = int(bounds[xx, 0]), int(bounds[xx, 1])
        k = kk[xx, :cnt].astype(np.int64).reshape((cnt,) + (1,) * (a.ndim - 1))
        ss = (1 << (PRECISION_BITS - 1)) + (a[xmin:xmin + cnt] * k).sum(0)
        out[xx] = np.clip(ss >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize(img: np.ndarray, size_wh: Tuple[int, int], filt: int) -> np.ndarray:
    """`PIL.Image.resize((w, h), resample)` for RGB: copy when the size is unchanged, else horizontal pass then vertical pass."""
    w, h = size_wh
    out = img
    if w != img.shape[1]:
        out = _resample_axis(out, w, filt, 1)
    if h != img.shape[0]:
        out = _resample_axis(out, h, filt, 0)
    return out.copy() if out is img else out


# --------------------------------------------------------------------------------------- Image.transform(AFFINE) (Geometry.c)
def _coord(v: float) -> int:
    return -1 if v < 0.0 else int(v)


def scale_affine_tables(a: List[float], in_w: int, in_h: int, out_w: int, out_h: int):
    """`ImagingScaleAffine`: per-column / per-row source indices (-1 = outside, zero-filled).  The offsets ACCUMULATE (xo += a[0])."""
    xt = np.full(out_w, -1, np.int32)
    yt = np.full(out_h, -1, np.int32)
    xo = a[2] + a[0] * 0.5
    for x in range(out_w):
        xin = _coord(xo)
        if 0 <= xin < in_w:
            xt[x] = xin
        xo += a[0]
    yo = a[5] + a[4] * 0.5
    for y in range(out_h):
        yin = _coord(yo)
        if 0 <= yin < in_h:
            yt[y] = yin
        yo += a[4]
    return xt, yt


def _bicubic_poly(v1, v2, v3, v4, d):
    p1 = v2
    p2 = -v1 + v3
    p3 = 2 * (v1 - v2) + v3 - v4
    p4 = -v1 + v2 - v3 + v4
    return p1 + d * (p2 + d * (p3 + d * p4))


def affine_transform(img: np.ndarray, a: List[float], filt: int, out_wh: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """`Image.transform(size, AFFINE, a, resample)` for an axis-aligned matrix (a[1] == a[3] == 0 -- the only kind the reference's ops build);
    pixels that map outside the source are zero-filled."""
    assert a[1] == 0 and a[3] == 0, "only scale + translate matrices occur on this path"
    H, W = img.shape[:2]
    ow, oh = out_wh if out_wh is not None else (W, H)
    out = np.zeros((oh, ow, 3), np.uint8)
    if filt == NEAREST:
        xt, yt = scale_affine_tables(a, W, H, ow, oh)
        # Geometry.c copies columns xmin..xmax of rows whose source row is inside
        vx, vy = xt >= 0, yt >= 0
        out[np.ix_(vy, vx)] = img[np.ix_(yt[vy], xt[vx])]
        return out
    assert filt == BICUBIC
    # affine_transform(): xin = a0 * (x + 0.5) + a1 * (y + 0.5) + a2 (a1 = 0 adds an exact zero)
    xs = a[0] * (np.arange(ow, dtype=np.float64) + 0.5) + a[1] * 0.5 + a[2]
    ys = a[3] * 0.5 + a[4] * (np.arange(oh, dtype=np.float64) + 0.5) + a[5]
    okx = (xs >= 0.0) & (xs < W)
    oky = (ys >= 0.0) & (ys < H)
    xin, yin = xs - 0.5, ys - 0.5
    x0 = np.floor(xin).astype(np.int64)
    y0 = np.floor(yin).astype(np.int64)
    dx = (xin - x0)[None, :, None]
    dy = (yin - y0)[:, None, None]
    x0 -= 1
    y0 -= 1
    cx = [np.clip(x0 + i, 0, W - 1) for i in range(4)]
    cy = [np.clip(y0 + i, 0, H - 1) for i in range(4)]  # Geometry.c's "else v_{i+1} = v_i" chain equals a clamp
    src = img.astype(np.float64)
    rows = []
    for i in range(4):
        r = src[cy[i]]
        rows.append(_bicubic_poly(r[:, cx[0]], r[:, cx[1]], r[:, cx[2]], r[:, cx[3]], dx))
    v = _bicubic_poly(rows[0], rows[1], rows[2], rows[3], dy)
    q = np.where(v <= 0.0, 0, np.where(v >= 255.0, 255, np.floor(np.clip(v, 0, 255)))).astype(np.uint8)  # (UINT8) v: truncation
    ok = oky[:, None] & okx[None, :]
    out[ok] = q[ok]
    return out


# ----------------------------------------------------------------------------------------------- torchvision glue (unpinned)
def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision `_get_inverse_affine_matrix(center, angle, translate, scale, shear, inverted=True)`."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def tv_affine(img, translate, scale, filt):
    """`v2.functional.affine(image, angle=0, translate, scale, shear=0, interpolation)` on a PIL image: centre = (w/2, h/2)."""
    H, W = img.shape[:2]
    m = inverse_affine_matrix([W * 0.5, H * 0.5], 0.0, [float(translate[0]), float(translate[1])], scale, [0.0, 0.0])
    return affine_transform(img, m, filt)


def tv_pad_edge(img, pad_lr: int, pad_tb: int):
    """`v2.functional.pad(image, (lr, tb), padding_mode="edge")`."""
    return np.pad(img, ((pad_tb, pad_tb), (pad_lr, pad_lr), (0, 0)), mode="edge")


def tv_center_crop(img, output_size):
    """`v2.functional.center_crop(image, (crop_h, crop_w))`: zero-pads first when the image is smaller, anchor = round(diff / 2)."""
    ch, cw = int(output_size[0]), int(output_size[1])
    H, W = img.shape[:2]
    if ch > H or cw > W:
        pl = (cw - W) // 2 if cw > W else 0
        pt = (ch - H) // 2 if ch > H else 0
        pr = (cw - W + 1) // 2 if cw > W else 0
        pb = (ch - H + 1) // 2 if ch > H else 0
        img = np.pad(img, ((pt, pb), (pl, pr), (0, 0)))
        H, W = img.shape[:2]
        if cw == W and ch == H:
            return img
    top = int(round((H - ch) / 2.0))
    left = int(round((W - cw) / 2.0))
    return img[top:top + ch, left:left + cw].copy()


def tv_resize_short_edge(img, size: int, filt=LANCZOS):
    """`v2.Resize(size, interpolation)` with an int size: short edge -> size, long edge -> int(size * long / short)."""
    H, W = img.shape[:2]
    short, long = (W, H) if W <= H else (H, W)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if W <= H else (new_long, new_short)
    return resize(img, (nw, nh), filt)


def to_pixel_values(img):
    """`ToImage` + `ToDtype(float32, scale=True)` + `Normalize(0.5, 0.5)`: fp32 CHW, fl(fl(fl(v) * fl(1/255)) - 0.5) / 0.5."""
    x = img.astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def grayscale(img):
    """`PIL.ImageOps.grayscale(image).convert("RGB")`: ITU-R 601-2 luma in 16.16 fixed point, replicated to three bands."""
    v = img.astype(np.uint32)
    l = ((v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)
    return np.repeat(l[..., None], 3, axis=2)


# ------------------------------------------------------------- the reference's ops (paired_augmentation.py), draw order kept
def _compute_padding(h, w, scale):  # :12-17
    return round(((h / scale) - h) / 2), round(((w / scale) - w) / 2)


def adjust_scale(img, prompt, inversion=False):  # :20-49
    scale_factor = np.random.uniform(0.34, 1.4)
    h, w = img.shape[1], img.shape[0]  # the reference unpacks PIL's (width, height) as `h, w`
    pad_h, pad_w = _compute_padding(h, w, scale_factor)
    if pad_h > 0 and pad_w > 0:
        img = tv_pad_edge(img, pad_w, pad_h)
    img = tv_affine(img, (0, 0), scale_factor, BICUBIC)
    img = tv_center_crop(img, (h, w))
    if inversion:
        add = "<zoom-out_0> <zoom-out_1>" if scale_factor < 0.6 else "<zoom-in_0> <zoom-in_1>" if scale_factor > 1.2 else ""
    else:
        if scale_factor <= 0.6:
            add = np.random.choice(["a far away ", "very small "])
        elif scale_factor >= 1.2:
            add = np.random.choice(["zoomed in ", "close up "])
        else:
            add = ""
    return img, add + prompt


def horizontal_flip(img, prompt, inversion=False):  # :79-92
    img = img[:, ::-1].copy()
    first = np.random.rand() < 0.5
    word = "<hflip>" if inversion else "horizontally flipped"
    if first:
        prompt = word + " " + prompt
    else:
        prompt = prompt + ", " + word
    return img, prompt


def horizontal_translate(img, prompt, inversion=False):  # :95-134
    shift_dir = np.random.randint(0, 2)
    w, h = img.shape[1], img.shape[0]
    shift_str = np.random.uniform(low=0.15, high=0.3)
    shift = int(shift_str * w)
    trans = [-shift, 0] if shift_dir == 0 else [shift, 0]
    if inversion:
        add = " <left_0> <left_1> <left_2>" if shift_dir == 0 else " <right_0> <right_0> <right_0>"
    else:
        add = " on the left" if shift_dir == 0 else " on the right"
    prompt = prompt + add
    img = tv_pad_edge(img, shift, 0)
    img = tv_affine(img, trans, 1, NEAREST)
    img = tv_center_crop(img, [w, h])
    return img, prompt


def grayscale_op(img, prompt, inversion=False):  # :163-174
    return grayscale(img), f"{prompt}, " + ("<grayscale_0> <grayscale_1>" if inversion else "grayscale")


def random_resized_crop(img, target_size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):  # :177-217
    width, height = img.shape[1], img.shape[0]
    area = width * height * pyrandom.uniform(*scale)
    aspect_ratio = pyrandom.uniform(*ratio)
    new_width = min(int(round(math.sqrt(area * aspect_ratio))), width)
    new_height = min(int(round(math.sqrt(area / aspect_ratio))), height)
    x = pyrandom.randint(0, width - new_width)
    y = pyrandom.randint(0, height - new_height)
    return resize(img[y:y + new_height, x:x + new_width], target_size, BICUBIC)


def crop_op(img, prompt, inversion=False):  # :220-233
    h, w = img.shape[1], img.shape[0]  # (width, height) unpacked as h, w; passed on as target (width=h, height=w) -> unchanged size
    img = random_resized_crop(img, (h, w), ratio=(1.0, 1.0))
    add = "<crop>" if inversion else "cropped"
    if np.random.random() < 0.5:
        prompt = f"{add} {prompt}"
    else:
        prompt = f"{prompt}, {add}"
    return img, prompt


def square_photo_collage(img, prompt, inversion=False):  # :253-277
    axis = np.random.randint(2, 4)
    w, h = img.shape[1], img.shape[0]
    grid_w, grid_h = w // axis, h // axis
    small = resize(img, (grid_h, grid_w), BICUBIC).copy()  # PIL size (width=grid_h, height=grid_w) -> array [grid_w, grid_h, 3]
    small[0, :] = 0
    small[-1, :] = 0
    small[:, 0] = 0
    small[:, -1] = 0
    grid = np.tile(small, (axis, axis, 1))
    prompt = ("<collage_0> <collage_1> " if inversion else "photo collage of ") + prompt
    return grid, prompt


class PairedAugmentation:
    """paired_augmentation.py:280-351, same constructor and draw order (np.random for the gates / op choice, `random` inside crop)."""

    def __init__(self, hflip="false", inversion=False, p=0.5, color_prob=0.2, augment_prompt=True, ops="object"):
        assert hflip.lower() in ("true", "false", "inversion")
        self.hflip = False
        self.inversion, self.p, self.color_prob, self.augment_prompt = inversion, p, color_prob, augment_prompt
        if ops == "object":
            self.geometric_ops = [adjust_scale, crop_op, horizontal_translate]
            self.color_ops = [grayscale_op]
            self.other_ops = [square_photo_collage]
        else:
            self.geometric_ops, self.color_ops, self.other_ops = [], [grayscale_op], []
        if hflip == "inversion":
            self.geometric_ops.append(horizontal_flip)
        elif hflip == "true":
            self.hflip = True

    def __call__(self, img, prompt):
        if self.hflip and np.random.rand() < 0.5:
            img = img[:, ::-1].copy()
        for ops, prob in ((self.geometric_ops, self.p), (self.other_ops, self.p), (self.color_ops, self.color_prob)):
            if len(ops) > 0 and np.random.rand() < prob:
                op = np.random.choice(ops)
                img, new_prompt = op(img, prompt, self.inversion)
                if self.augment_prompt:
                    prompt = new_prompt
        return img, prompt, None


def dataset_item(img, instance_token, templates, size, center_crop, augment_pipe=None):
    """textboost/dataset.py:353-381 for one instance image: template draw (`random.randint`), augmentation, Resize(size, LANCZOS), crop
    (centre, or `RandomCrop.get_params` = two `torch.randint` draws), ToImage/ToDtype/Normalize.  Returns (pixel_values[3,size,size], prompt)."""
    import torch

    prompt_idx = pyrandom.randint(0, len(templates) - 1)
    prompt = templates[prompt_idx].format(instance_token)
    if augment_pipe is not None:
        img, prompt, _ = augment_pipe(img, prompt)
    img = tv_resize_short_edge(img, size, LANCZOS)
    H, W = img.shape[:2]
    if center_crop:
        img = tv_center_crop(img, (size, size))
    else:
        if W == size and H == size:
            i, j = 0, 0
        else:
            i = int(torch.randint(0, H - size + 1, size=(1,)).item())
            j = int(torch.randint(0, W - size + 1, size=(1,)).item())
        img = img[i:i + size, j:j + size]
    return to_pixel_values(img), prompt
