"""Device-side image augmentation + tokenised-prompt feeder.

Host mirror of the reference's data path for the instance images, with the same names, arguments, random-draw order and prompt edits:

    textboost/augment/paired_augmentation.py:20-351   adjust_scale, crop, horizontal_translate, horizontal_flip, grayscale,
                                                      square_photo_collage, PairedAugmentation
    textboost/dataset.py:324-381, :420-457            Resize(size, LANCZOS) -> Center/RandomCrop -> ToImage/ToDtype/Normalize, the template
                                                      draw, tokenize_prompt, collate_fn's {"input_ids", "pixel_values"}

The reference does this per sample in Pillow on DataLoader workers (8 x 512^2 Lanczos resizes per step at batch 8); here the few instance images
of a DreamBooth run are decoded once, stay resident in HBM as RGBX u8, and every op is a HIP kernel of csrc/image.hip that is bit-exact with
Pillow.  The host only draws the random parameters (np.random / random / torch, in the reference's order, so equal seeds give equal samples)
and builds O(W + H) index / coefficient tables.  There is no CPU fallback: a missing library raises.

An image is an int32 CUDA tensor [H, W] (one RGBX pixel per element).
"""
from __future__ import annotations

import collections
import math
import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops

NEAREST, LANCZOS, BICUBIC = 0, L.IMG_LANCZOS, L.IMG_BICUBIC

# resample tables per (in, out, filter, device): an LRU, because `crop` draws a new source size for almost every sample
_coeff_cache: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_COEFF_CACHE_MAX = 1024


def _dev_table(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def to_device_image(rgb) -> torch.Tensor:
    """uint8 [H, W, 3] (numpy / CPU or CUDA tensor; e.g. `np.asarray(PIL.Image.open(p).convert("RGB"))`) -> resident RGBX image."""
    t = torch.from_numpy(np.array(rgb, copy=True)) if isinstance(rgb, np.ndarray) else torch.as_tensor(rgb)
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("expected a uint8 [H, W, 3] RGB image")
    return ops.img_pack_rgb(t.cuda().contiguous())


def to_host_rgb(img: torch.Tensor) -> np.ndarray:
    return ops.img_unpack_rgb(img).cpu().numpy()


def resize(img: torch.Tensor, size_wh: Tuple[int, int], filt: int) -> torch.Tensor:
    """`PIL.Image.resize((w, h), resample)`: horizontal pass then vertical pass; a pass whose size does not change is skipped."""
    w, h = int(size_wh[0]), int(size_wh[1])
    H, W = img.shape
    out = img
    for vertical, n_in, n_out in ((0, W, w), (1, H, h)):
        if n_in == n_out:
            continue
        key = (n_in, n_out, filt, img.device)
        if key not in _coeff_cache:
            _, bounds, kk = ops.resample_coeffs(n_in, n_out, filt)
            _coeff_cache[key] = (bounds.to(img.device), kk.to(img.device), ops.resample_fit24(kk))
            if len(_coeff_cache) > _COEFF_CACHE_MAX:
                _coeff_cache.popitem(last=False)
        else:
            _coeff_cache.move_to_end(key)
        bounds, kk, fit24 = _coeff_cache[key]
        out = ops.img_resample(out, n_out, bounds, kk, vertical, fit24)
    return out.clone() if out is img else out


def _gather(img, xt: np.ndarray, yt: np.ndarray, gray=False):
    return ops.img_gather(img, _dev_table(xt, img.device), _dev_table(yt, img.device), gray)


def _center_crop_offsets(W: int, H: int, crop_h: int, crop_w: int):
    """torchvision `center_crop(image, (crop_h, crop_w))` on a W x H image: output pixel (x, y) is source pixel (x + ox, y + oy), zero where
    that falls outside (the zero padding it applies first when the image is smaller than the crop)."""
    pl = (crop_w - W) // 2 if crop_w > W else 0
    pt = (crop_h - H) // 2 if crop_h > H else 0
    pr = (crop_w - W + 1) // 2 if crop_w > W else 0
    pb = (crop_h - H + 1) // 2 if crop_h > H else 0
    left = int(round((W + pl + pr - crop_w) / 2.0))
    top = int(round((H + pt + pb - crop_h) / 2.0))
    return left - pl, top - pt


def _center_crop_tables(W: int, H: int, crop_h: int, crop_w: int):
    ox, oy = _center_crop_offsets(W, H, crop_h, crop_w)
    xs, ys = np.arange(crop_w) + ox, np.arange(crop_h) + oy
    return np.where((xs >= 0) & (xs < W), xs, -1), np.where((ys >= 0) & (ys < H), ys, -1)


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision `_get_inverse_affine_matrix` (what `v2.functional.affine` hands to `PIL.Image.transform`)."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [v / scale for v in [d, -b, 0.0, -c, a, 0.0]]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


# ------------------------------------------------------------------------------------ the reference's ops, draw order kept
def _compute_padding(h, w, scale):
    return round(((h / scale) - h) / 2), round(((w / scale) - w) / 2)


def adjust_scale(image, prompt, inversion=False):
    """paired_augmentation.py:20-49 -- pad(edge) + affine(scale, BICUBIC) + center_crop in ONE launch."""
    scale_factor = np.random.uniform(0.34, 1.4)
    H, W = image.shape
    h, w = W, H  # the reference unpacks PIL's (width, height) as `h, w`
    pad_h, pad_w = _compute_padding(h, w, scale_factor)
    px, py = (pad_w, pad_h) if (pad_h > 0 and pad_w > 0) else (0, 0)  # v2.functional.pad(image, (pad_w, pad_h)): left/right, top/bottom
    PW, PH = W + 2 * px, H + 2 * py
    m = inverse_affine_matrix([PW * 0.5, PH * 0.5], 0.0, [0.0, 0.0], scale_factor, [0.0, 0.0])
    crop_h, crop_w = h, w  # center_crop(image, (h, w))
    ox, oy = _center_crop_offsets(PW, PH, crop_h, crop_w)
    image = ops.img_affine_bicubic(image, m, px, py, ox, oy, crop_w, crop_h)
    if inversion:
        add = "<zoom-out_0> <zoom-out_1>" if scale_factor < 0.6 else "<zoom-in_0> <zoom-in_1>" if scale_factor > 1.2 else ""
    else:
        if scale_factor <= 0.6:
            add = np.random.choice(["a far away ", "very small "])
        elif scale_factor >= 1.2:
            add = np.random.choice(["zoomed in ", "close up "])
        else:
            add = ""
    return image, add + prompt


def horizontal_flip(image, prompt, inversion=False):
    """paired_augmentation.py:79-92."""
    H, W = image.shape
    image = _gather(image, np.arange(W - 1, -1, -1), np.arange(H))
    word = "<hflip>" if inversion else "horizontally flipped"
    if np.random.rand() < 0.5:
        prompt = word + " " + prompt
    else:
        prompt = prompt + ", " + word
    return image, prompt


def horizontal_translate(image, prompt, inversion=False):
    """paired_augmentation.py:95-134 -- pad(edge) + affine(translate, NEAREST) + center_crop composed into one index map, one launch."""
    shift_dir = np.random.randint(0, 2)
    H, W = image.shape
    w, h = W, H
    shift_str = np.random.uniform(low=0.15, high=0.3)
    shift = int(shift_str * w)
    trans = [-shift, 0] if shift_dir == 0 else [shift, 0]
    if inversion:
        add = " <left_0> <left_1> <left_2>" if shift_dir == 0 else " <right_0> <right_0> <right_0>"
    else:
        add = " on the left" if shift_dir == 0 else " on the right"
    prompt = prompt + add
    PW = W + 2 * shift
    pad_x = np.clip(np.arange(-shift, W + shift), 0, W - 1)  # padded column -> source column
    m = inverse_affine_matrix([PW * 0.5, H * 0.5], 0.0, [float(trans[0]), float(trans[1])], 1, [0.0, 0.0])
    ax, ay = ops.affine_nearest_tables(m, PW, H, PW, H)
    ax, ay = ax.numpy(), ay.numpy()
    cx, cy = _center_crop_tables(PW, H, w, h)  # center_crop(image, [w, h]): crop_h = w, crop_w = h
    fx = np.where(cx >= 0, ax[np.maximum(cx, 0)], -1)
    fy = np.where(cy >= 0, ay[np.maximum(cy, 0)], -1)
    fx = np.where(fx >= 0, pad_x[np.maximum(fx, 0)], -1)
    return _gather(image, fx, fy), prompt


def grayscale(image, prompt, inversion=False, size=None):
    """paired_augmentation.py:163-174."""
    H, W = image.shape
    image = _gather(image, np.arange(W), np.arange(H), gray=True)
    return image, f"{prompt}, " + ("<grayscale_0> <grayscale_1>" if inversion else "grayscale")


def random_resized_crop(image, target_size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
    """paired_augmentation.py:177-217 (`random` module draws); the crop is a strided view, only the BICUBIC resize touches memory."""
    height, width = image.shape
    area = width * height * random.uniform(*scale)
    aspect_ratio = random.uniform(*ratio)
    new_width = min(int(round(math.sqrt(area * aspect_ratio))), width)
    new_height = min(int(round(math.sqrt(area / aspect_ratio))), height)
    x = random.randint(0, width - new_width)
    y = random.randint(0, height - new_height)
    return resize(image[y:y + new_height, x:x + new_width], target_size, BICUBIC)


def crop(image, prompt, inversion=False):
    """paired_augmentation.py:220-233."""
    H, W = image.shape
    image = random_resized_crop(image, (W, H), ratio=(1.0, 1.0))
    add = "<crop>" if inversion else "cropped"
    if np.random.random() < 0.5:
        prompt = f"{add} {prompt}"
    else:
        prompt = f"{prompt}, {add}"
    return image, prompt


def square_photo_collage(image, prompt, inversion=False):
    """paired_augmentation.py:253-277: BICUBIC resize to one cell, black 1-pixel frame, axis x axis tiling (frame + tiling = one index map)."""
    axis = np.random.randint(2, 4)
    H, W = image.shape
    grid_w, grid_h = W // axis, H // axis
    small = resize(image, (grid_h, grid_w), BICUBIC)  # PIL size (width = grid_h, height = grid_w)
    sh, sw = small.shape
    xs, ys = np.arange(sw), np.arange(sh)
    xs[[0, -1]] = -1
    ys[[0, -1]] = -1
    image = _gather(small, np.tile(xs, axis), np.tile(ys, axis))
    prompt = ("<collage_0> <collage_1> " if inversion else "photo collage of ") + prompt
    return image, prompt


class PairedAugmentation:
    """paired_augmentation.py:280-351: same constructor, same gates and op choice (np.random), returns (image, prompt, None)."""

    def __init__(self, hflip="false", inversion=False, p=0.5, color_prob=0.2, augment_prompt=True, ops="object"):
        assert hflip.lower() in ("true", "false", "inversion"), f"Invalid hflip value: {hflip}"
        self.hflip = False
        self.inversion = inversion
        self.p = p
        self.color_prob = color_prob
        self.augment_prompt = augment_prompt
        if ops == "object":
            self.geometric_ops = [adjust_scale, crop, horizontal_translate]
            self.color_ops = [grayscale]
            self.other_ops = [square_photo_collage]
        else:  # "style"
            self.geometric_ops = []
            self.color_ops = [grayscale]
            self.other_ops = []
        if hflip == "inversion":
            self.geometric_ops.append(horizontal_flip)
        elif hflip == "true":
            self.hflip = True

    def __call__(self, image, prompt):
        if not (isinstance(image, torch.Tensor) and image.is_cuda and image.dtype == torch.int32 and image.dim() == 2):
            raise TypeError("PairedAugmentation takes a resident RGBX image (int32 CUDA tensor [H, W]); see to_device_image()")
        if self.hflip and np.random.rand() < 0.5:
            H, W = image.shape
            image = _gather(image, np.arange(W - 1, -1, -1), np.arange(H))
        for op_list, prob in ((self.geometric_ops, self.p), (self.other_ops, self.p), (self.color_ops, self.color_prob)):
            if len(op_list) > 0 and np.random.rand() < prob:
                op = np.random.choice(op_list)
                image, new_prompt = op(image, prompt, self.inversion)
                if self.augment_prompt:
                    prompt = new_prompt
        return image, prompt, None


# ------------------------------------------------------------------------------------------------------------------ the feeder
def resize_short_edge(image, size: int, filt=LANCZOS):
    """`v2.Resize(size, interpolation=LANCZOS)` with an int size (dataset.py:324)."""
    H, W = image.shape
    short, long = (W, H) if W <= H else (H, W)
    new_long = int(size * long / short)
    nw, nh = (size, new_long) if W <= H else (new_long, size)
    return resize(image, (nw, nh), filt)


class PromptFeeder:
    """`tokenize_prompt` (dataset.py:79-93) behind a cache: the prompt space of a run is the templates x the augmentation captions, so after a few
    steps every prompt is a dictionary hit and the tokenizer leaves the step."""

    def __init__(self, tokenizer, max_length: Optional[int] = None):
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.cache: Dict[str, torch.Tensor] = {}
        self.hits = 0

    def __call__(self, prompt: str) -> torch.Tensor:
        ids = self.cache.get(prompt)
        if ids is None:
            max_length = self.max_length if self.max_length is not None else self.tokenizer.model_max_length
            out = self.tokenizer(prompt, truncation=True, padding="max_length", max_length=max_length, return_tensors="pt")
            ids = self.cache[prompt] = out.input_ids
        else:
            self.hits += 1
        return ids


class DeviceFeeder:
    """`TextBoostDataset.__getitem__` + `collate_fn` (dataset.py:353-381, :420-457) for the instance images, on the device.

    images: list of (resident RGBX image, instance_token).  `batch(indices)` returns {"input_ids": int64 [B, 77] (CPU, as the reference's
    collate_fn), "pixel_values": fp32 [B, 3, size, size] on the device, "prompts": list[str]}."""

    def __init__(self, images: Sequence[Tuple[torch.Tensor, str]], tokenizer, templates: Sequence[str], size=512, center_crop=False,
                 augment_pipe: Optional[PairedAugmentation] = None):
        if len(images) == 0:
            raise ValueError("no instance images")
        self.images = list(images)
        self.templates = list(templates)
        self.size = size
        self.center_crop = center_crop
        self.augment_pipe = augment_pipe
        self.tokenize = PromptFeeder(tokenizer)

    def __len__(self):
        return len(self.images)

    def item_into(self, index: int, dst: torch.Tensor):
        """One sample; pixel_values written into dst [3, size, size].  Returns (input_ids [1, 77], prompt)."""
        image, instance_token = self.images[index % len(self.images)]
        prompt_idx = random.randint(0, len(self.templates) - 1)
        prompt = self.templates[prompt_idx].format(instance_token)
        if self.augment_pipe is not None:
            image, prompt, _ = self.augment_pipe(image, prompt)
        image = resize_short_edge(image, self.size, LANCZOS)
        H, W = image.shape
        if self.center_crop:
            y1 = max(0, int(round((H - self.size) / 2.0)))
            x1 = max(0, int(round((W - self.size) / 2.0)))
        elif W == self.size and H == self.size:
            y1, x1 = 0, 0
        else:  # v2.RandomCrop.get_params: two torch.randint draws from the global generator
            y1 = int(torch.randint(0, H - self.size + 1, size=(1,)).item())
            x1 = int(torch.randint(0, W - self.size + 1, size=(1,)).item())
        ops.img_to_pixels(image, x1, y1, dst)
        return self.tokenize(prompt), prompt

    def batch(self, indices: Sequence[int], out: Optional[torch.Tensor] = None):
        B = len(indices)
        if out is None:
            out = torch.empty(B, 3, self.size, self.size, dtype=torch.float32, device=self.images[0][0].device)
        ids, prompts = [], []
        for b, i in enumerate(indices):
            t, p = self.item_into(i, out[b])
            ids.append(t)
            prompts.append(p)
        return {"input_ids": torch.cat(ids, dim=0), "pixel_values": out, "prompts": prompts}


class PrefetchFeeder:
    """Overlaps the feeder with the training step.  `DeviceFeeder.batch` is host-bound (random draws, O(W + H) table uploads, a handful of
    launches per sample: ~0.3 ms / sample with augmentation) and its uploads are stream-ordered, so on the step's stream the host would sit
    behind the running step.  Here batch k+1 is produced on a side stream into a staging buffer while step k runs; `commit()` makes the step's
    stream wait for it and copies pixels / ids into the step's static inputs (one device-to-device copy)."""

    def __init__(self, feeder: DeviceFeeder, batch: int, pixel_values: torch.Tensor, input_ids: torch.Tensor):
        self.feeder, self.batch = feeder, batch
        self.pixel_values, self.input_ids = pixel_values, input_ids
        self.staging = torch.empty_like(pixel_values)
        self.ids_staging = torch.empty_like(input_ids)
        self.side = torch.cuda.Stream(device=pixel_values.device)
        self.ready = torch.cuda.Event()
        self.consumed = None  # recorded on the step's stream right after a commit's copies: the staging buffers are free from there on
        self.pending = None

    def prefetch(self, indices: Sequence[int]):
        if self.consumed is not None:
            self.side.wait_event(self.consumed)  # NOT wait_stream(main): that would put the side stream (and the host's uploads) behind the step
        with torch.cuda.stream(self.side):
            b = self.feeder.batch(indices, out=self.staging)
            self.ids_staging.copy_(b["input_ids"], non_blocking=True)
            self.ready.record(self.side)
        self.pending = b

    def commit(self):
        """-> the prompts of the committed batch."""
        if self.pending is None:
            raise RuntimeError("commit() without a prefetch()")
        torch.cuda.current_stream().wait_event(self.ready)
        self.pixel_values.copy_(self.staging)
        self.input_ids.copy_(self.ids_staging)
        self.consumed = torch.cuda.Event()
        self.consumed.record(torch.cuda.current_stream())
        b, self.pending = self.pending, None
        return b["prompts"]
