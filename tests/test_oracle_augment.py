"""CPU tests: the augmentation oracle (oracle/augment.py) against the committed fixtures -- Pillow's own results for the primitives and
seeded runs of the reference's `PairedAugmentation` (tests/golden/gen_augment_golden.py) -- bit-exact: this is byte work."""
import json
import os
import random
import zlib

import numpy as np
import pytest

from oracle import augment as A

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def prim():
    return np.load(os.path.join(G, "augment_pil_primitives.npz"))


def test_resize_matches_pillow(prim):
    for i, (h, w, oh, ow) in enumerate(prim["resize_cases"]):
        a = prim[f"resize_in_{i}"]
        for name, f in (("bicubic", A.BICUBIC), ("lanczos", A.LANCZOS)):
            got = A.resize(a, (int(ow), int(oh)), f)
            assert np.array_equal(got, prim[f"resize_{name}_{i}"]), (i, name)


def test_affine_matches_pillow(prim):
    for i, (h, w) in enumerate(prim["affine_cases"]):
        a, m = prim[f"affine_in_{i}"], [float(v) for v in prim[f"affine_matrix_{i}"]]
        assert np.array_equal(A.affine_transform(a, m, A.NEAREST), prim[f"affine_nearest_{i}"]), i
        assert np.array_equal(A.affine_transform(a, m, A.BICUBIC), prim[f"affine_bicubic_{i}"]), i


def test_grayscale_matches_pillow(prim):
    assert np.array_equal(A.grayscale(prim["gray_in"]), prim["gray_out"])


def test_live_pillow_if_present():
    """Whenever Pillow imports (it does in this image), re-check a fresh seeded sweep live, incl. the sizes the dataset really uses."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    r = np.random.default_rng(5)
    for _ in range(6):
        h, w = int(r.integers(8, 200)), int(r.integers(8, 200))
        oh, ow = int(r.integers(4, 160)), int(r.integers(4, 160))
        a = r.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for f, pf in ((A.BICUBIC, Image.BICUBIC), (A.LANCZOS, Image.LANCZOS)):
            assert np.array_equal(A.resize(a, (ow, oh), f), np.asarray(Image.fromarray(a).resize((ow, oh), pf)))
    a = r.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    got = A.tv_resize_short_edge(a, 128)
    assert got.shape == (128, 179, 3)
    assert np.array_equal(got, np.asarray(Image.fromarray(a).resize((179, 128), Image.LANCZOS)))


def test_paired_augmentation_matches_reference_runs():
    """Same seeds -> same random draws, same prompt edits, same pixels as the real paired_augmentation.py."""
    meta = json.load(open(os.path.join(G, "augment_reference_calls.json")))
    imgs = np.load(os.path.join(G, "augment_reference_images.npz"))
    kinds = set()
    for r in meta["records"]:
        pipe = A.PairedAugmentation(**r["config"])
        np.random.seed(r["np_seed"])
        random.seed(r["py_seed"])
        out, prompt, mask = pipe(imgs[f"in_{r['case']}"], meta["prompt_in"])
        assert mask is None
        assert prompt == r["prompt"], r["case"]
        assert list(out.shape[:2]) == r["out_hw"], r["case"]
        assert np.array_equal(out, imgs[f"out_{r['case']}"]), (r["case"], r["calls"])
        assert zlib.crc32(np.ascontiguousarray(out).tobytes()) == r["crc32"]
        kinds.add("+".join(c[0] for c in r["calls"]))
    assert {"", "affine+center_crop", "pad+affine+center_crop"} <= kinds


def test_glue_arguments_match_what_the_reference_passed():
    """The recorded v2.functional arguments (pad widths, scale, translate, crop size) are what the oracle derives from the same draws."""
    meta = json.load(open(os.path.join(G, "augment_reference_calls.json")))
    seen = 0
    for r in meta["records"]:
        calls = r["calls"]
        if not calls or r["config"]["ops"] != "object":
            continue
        aff = [c for c in calls if c[0] == "affine"][0]
        if aff[5] == A.BICUBIC:  # adjust_scale: scale in [0.34, 1.4], padding by _compute_padding on the (w, h)-swapped names
            h, w = r["in_hw"][1], r["in_hw"][0]
            pad_h, pad_w = A._compute_padding(h, w, aff[3])
            pads = [c for c in calls if c[0] == "pad"]
            assert (len(pads) == 1) == (pad_h > 0 and pad_w > 0)
            if pads:
                assert pads[0][1] == [pad_w, pad_h]
            seen += 1
        else:  # horizontal_translate: nearest, integer shift, same pad on both sides
            pads = [c for c in calls if c[0] == "pad"][0]
            assert abs(aff[2][0]) == pads[1][0] and aff[3] == 1.0
            seen += 1
    assert seen >= 10


def test_to_pixel_values_and_dataset_item():
    import torch
    a = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, 2)
    pv = A.to_pixel_values(a)
    ref = (torch.from_numpy(a).permute(2, 0, 1).to(torch.float32).mul_(1.0 / 255)).sub_(0.5).div_(0.5)
    assert np.array_equal(pv, ref.numpy())
    assert pv.min() == -1.0 and pv.max() == 1.0
    img = np.random.default_rng(3).integers(0, 256, (90, 70, 3), dtype=np.uint8)
    random.seed(1), np.random.seed(1), torch.manual_seed(1)
    pv, prompt = A.dataset_item(img, "<sks> dog", ["a photo of {}", "{} in a bucket"], 32, False, A.PairedAugmentation(inversion=True, hflip="inversion"))
    assert pv.shape == (3, 32, 32) and pv.dtype == np.float32 and "<sks> dog" in prompt
    random.seed(1), np.random.seed(1), torch.manual_seed(1)
    pv2, prompt2 = A.dataset_item(img, "<sks> dog", ["a photo of {}", "{} in a bucket"], 32, False, A.PairedAugmentation(inversion=True, hflip="inversion"))
    assert np.array_equal(pv, pv2) and prompt == prompt2


def test_center_crop_pads_when_smaller():
    a = np.full((10, 6, 3), 9, np.uint8)
    out = A.tv_center_crop(a, (6, 10))  # the reference's swapped (w, h) on a non-square image
    assert out.shape == (6, 10, 3) and out[:, :2].max() == 0 and out[:, 2:8].min() == 9 and out[:, 8:].max() == 0


def test_c_abi_host_table_builders_match_oracle():
    """tb_resample_coeffs / tb_affine_nearest_tables are HOST entry points (libm sin, sequential accumulation): checkable without a GPU."""
    from textboost_amd import ops
    for n, o in [(53, 16), (64, 32), (30, 91), (257, 64), (7, 2), (5, 40), (2048, 512), (1365, 512), (512, 513), (1, 4)]:
        for f in (A.LANCZOS, A.BICUBIC):
            ks, b, kk = ops.resample_coeffs(n, o, f)
            ks2, b2, kk2 = A.precompute_coeffs(n, 0.0, float(n), o, f)
            assert ks == ks2 and ks % 2 == 1 and np.array_equal(b.numpy(), b2) and np.array_equal(kk.numpy(), kk2), (n, o, f)
    for s, t, w, h in [(0.34, 0, 40, 40), (0.77, 0, 50, 33), (1.4, 0, 48, 64), (1.0, -7, 40, 40), (1.0, 13, 50, 33), (0.9, 3.5, 64, 48)]:
        m = A.inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [float(t), 0.0], s, [0.0, 0.0])
        xt, yt = ops.affine_nearest_tables(m, w, h, w, h)
        xt2, yt2 = A.scale_affine_tables(m, w, h, w, h)
        assert np.array_equal(xt.numpy(), xt2) and np.array_equal(yt.numpy(), yt2)
    with pytest.raises(RuntimeError):
        ops.affine_nearest_tables([1.0, 0.1, 0.0, 0.0, 1.0, 0.0], 8, 8, 8, 8)
