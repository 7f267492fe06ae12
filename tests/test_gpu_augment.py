"""Device-side augmentation + feeder on the GPU, through the C-ABI, BIT-EXACT against the numpy oracle, against Pillow's
committed outputs and against the committed seeded runs of the reference's own PairedAugmentation (tests/golden/gen_augment_golden.py)."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rnd(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def up(a):
    from textboost_amd import augment as D
    return D.to_device_image(a)


def down(t):
    from textboost_amd import augment as D
    return D.to_host_rgb(t)


def test_pack_unpack_roundtrip_and_strided_view():
    a = rnd(0, 37, 53)
    t = up(a)
    assert t.dtype == torch.int32 and tuple(t.shape) == (37, 53)
    assert np.array_equal(down(t), a)
    assert np.array_equal(down(t[5:30, 7:40]), a[5:30, 7:40])


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 16, 16), (64, 64, 64, 32), (100, 80, 33, 47), (30, 30, 77, 91), (257, 129, 64, 64), (9, 7, 3, 2),
                                         (5, 5, 40, 40), (1, 1, 4, 4), (200, 300, 64, 96), (70, 1000, 70, 65), (640, 3, 17, 3)])
def test_resize_matches_oracle(h, w, oh, ow):
    from oracle import augment as A
    from textboost_amd import augment as D
    a = rnd(h * 1000 + w, h, w)
    for fo, fd in ((A.BICUBIC, D.BICUBIC), (A.LANCZOS, D.LANCZOS)):
        assert np.array_equal(down(D.resize(up(a), (ow, oh), fd)), A.resize(a, (ow, oh), fo)), (fo,)


def test_resize_matches_pillow_fixture():
    from textboost_amd import augment as D
    prim = np.load(os.path.join(G, "augment_pil_primitives.npz"))
    for i, (h, w, oh, ow) in enumerate(prim["resize_cases"]):
        t = up(prim[f"resize_in_{i}"])
        assert np.array_equal(down(D.resize(t, (int(ow), int(oh)), D.BICUBIC)), prim[f"resize_bicubic_{i}"]), i
        assert np.array_equal(down(D.resize(t, (int(ow), int(oh)), D.LANCZOS)), prim[f"resize_lanczos_{i}"]), i


def test_resize_of_a_strided_crop_and_same_size_copy():
    from oracle import augment as A
    from textboost_amd import augment as D
    a = rnd(3, 90, 120)
    t = up(a)
    got = D.resize(t[10:70, 20:100], (64, 48), D.BICUBIC)
    assert np.array_equal(down(got), A.resize(a[10:70, 20:100], (64, 48), A.BICUBIC))
    same = D.resize(t, (120, 90), D.LANCZOS)
    assert same.data_ptr() != t.data_ptr() and np.array_equal(down(same), a)


def test_full_size_lanczos_like_the_dataset():
    """A camera-sized instance image through `v2.Resize(512, LANCZOS)` (dataset.py:324): 1536 x 2048 -> 512 x 682, both passes > 2.6x down."""
    from oracle import augment as A
    from textboost_amd import augment as D
    r = np.random.default_rng(11)
    yy, xx = np.mgrid[0:1536, 0:2048]
    a = np.clip(np.stack([xx // 8, yy // 6, (xx + yy) // 14], -1) + r.integers(-30, 31, (1536, 2048, 3)), 0, 255).astype(np.uint8)
    got = down(D.resize_short_edge(up(a), 512))
    assert got.shape == (512, 682, 3)
    assert np.array_equal(got, A.tv_resize_short_edge(a, 512))


def test_affine_matches_pillow_fixture():
    from textboost_amd import augment as D
    from textboost_amd import ops
    prim = np.load(os.path.join(G, "augment_pil_primitives.npz"))
    for i, (h, w) in enumerate(prim["affine_cases"]):
        h, w = int(h), int(w)
        t, m = up(prim[f"affine_in_{i}"]), [float(v) for v in prim[f"affine_matrix_{i}"]]
        got = ops.img_affine_bicubic(t, m, 0, 0, 0, 0, w, h)
        assert np.array_equal(down(got), prim[f"affine_bicubic_{i}"]), i
        xt, yt = ops.affine_nearest_tables(m, w, h, w, h)
        got = ops.img_gather(t, xt.cuda(), yt.cuda())
        assert np.array_equal(down(got), prim[f"affine_nearest_{i}"]), i


@pytest.mark.parametrize("h,w", [(40, 40), (33, 50), (64, 48), (131, 77)])
def test_fused_pad_affine_crop_matches_the_three_step_oracle(h, w):
    from oracle import augment as A
    from textboost_amd import augment as D
    a = rnd(h + w, h, w)
    for seed in range(6):
        np.random.seed(seed)
        want, p1 = A.adjust_scale(a, "x", True)
        np.random.seed(seed)
        got, p2 = D.adjust_scale(up(a), "x", True)
        assert p1 == p2 and np.array_equal(down(got), want), (h, w, seed)


def test_gather_ops_match_oracle():
    from oracle import augment as A
    from textboost_amd import augment as D
    for (h, w) in [(40, 40), (33, 50), (64, 48)]:
        a = rnd(7 * h + w, h, w)
        for name in ("horizontal_flip", "horizontal_translate", "square_photo_collage", "crop"):
            fo = A.crop_op if name == "crop" else getattr(A, name)
            for seed in range(5):
                np.random.seed(seed), random.seed(seed)
                want, p1 = fo(a, "a photo", False)
                np.random.seed(seed), random.seed(seed)
                got, p2 = getattr(D, name)(up(a), "a photo", False)
                assert p1 == p2, (name, seed)
                assert np.array_equal(down(got), want), (name, h, w, seed)
        want, _ = A.grayscale_op(a, "p")
        got, _ = D.grayscale(up(a), "p")
        assert np.array_equal(down(got), want)


def test_paired_augmentation_matches_the_reference_runs():
    """Same seeds as the recorded runs of the real paired_augmentation.py -> same prompts, same pixels, for all 40 cases."""
    from textboost_amd import augment as D
    meta = json.load(open(os.path.join(G, "augment_reference_calls.json")))
    imgs = np.load(os.path.join(G, "augment_reference_images.npz"))
    for r in meta["records"]:
        pipe = D.PairedAugmentation(**r["config"])
        np.random.seed(r["np_seed"])
        random.seed(r["py_seed"])
        out, prompt, mask = pipe(up(imgs[f"in_{r['case']}"]), meta["prompt_in"])
        assert mask is None and prompt == r["prompt"], r["case"]
        assert np.array_equal(down(out), imgs[f"out_{r['case']}"]), (r["case"], r["calls"])


class FakeTokenizer:
    """Deterministic stand-in (no tokenizer files exist offline): ids from a hash of the words, padded to 77 like CLIPTokenizer."""
    model_max_length = 77

    def __init__(self):
        self.calls = 0

    def __call__(self, prompt, truncation=True, padding="max_length", max_length=77, return_tensors="pt"):
        self.calls += 1
        ids = [49406] + [sum(map(ord, w)) % 49405 for w in prompt.split()][:max_length - 2]
        ids = ids + [49407] * (max_length - len(ids))
        return types.SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.int64))


def test_feeder_batch_matches_oracle_items():
    from oracle import augment as A
    from textboost_amd import augment as D
    imgs = [rnd(21, 150, 110), rnd(22, 96, 140)]
    templates = ["a photo of {}", "a rendering of {}", "{} on a table"]
    size = 64
    tok = FakeTokenizer()
    feeder = D.DeviceFeeder([(up(a), "<sks> dog") for a in imgs], tok, templates, size=size, center_crop=False,
                            augment_pipe=D.PairedAugmentation(hflip="inversion", inversion=True, p=0.8, color_prob=0.3))
    idx = [0, 1, 0, 1, 1, 0, 0, 1]
    random.seed(5), np.random.seed(5), torch.manual_seed(5)
    batch = feeder.batch(idx)
    random.seed(5), np.random.seed(5), torch.manual_seed(5)
    pipe = A.PairedAugmentation(hflip="inversion", inversion=True, p=0.8, color_prob=0.3)
    want = [A.dataset_item(imgs[i % 2], "<sks> dog", templates, size, False, pipe) for i in idx]
    assert batch["pixel_values"].shape == (8, 3, size, size) and batch["pixel_values"].dtype == torch.float32 and batch["pixel_values"].is_cuda
    assert batch["input_ids"].shape == (8, 77) and batch["input_ids"].dtype == torch.int64
    for b, (pv, prompt) in enumerate(want):
        assert batch["prompts"][b] == prompt
        assert np.array_equal(batch["pixel_values"][b].cpu().numpy(), pv), b
        assert torch.equal(batch["input_ids"][b:b + 1], FakeTokenizer()(prompt).input_ids)
    # the prompt cache: a second pass over the same draws never reaches the tokenizer
    calls = tok.calls
    random.seed(5), np.random.seed(5), torch.manual_seed(5)
    again = feeder.batch(idx)
    assert tok.calls == calls and feeder.tokenize.hits >= 8
    assert torch.equal(again["pixel_values"], batch["pixel_values"]) and torch.equal(again["input_ids"], batch["input_ids"])


def test_feeder_center_crop_and_vae_input_range():
    from oracle import augment as A
    from textboost_amd import augment as D
    a = rnd(31, 100, 260)
    feeder = D.DeviceFeeder([(up(a), "<v>")], FakeTokenizer(), ["{}"], size=48, center_crop=True, augment_pipe=None)
    random.seed(0)
    b = feeder.batch([0])
    random.seed(0)
    pv, _ = A.dataset_item(a, "<v>", ["{}"], 48, True, None)
    assert np.array_equal(b["pixel_values"][0].cpu().numpy(), pv)
    assert b["pixel_values"].min() >= -1 and b["pixel_values"].max() <= 1


def test_error_behaviour():
    from textboost_amd import _lib as L
    from textboost_amd import augment as D
    from textboost_amd import ops
    t = up(rnd(1, 16, 16))
    with pytest.raises(TypeError):
        D.PairedAugmentation()(np.zeros((4, 4, 3), np.uint8), "p")
    with pytest.raises(ValueError):
        D.to_device_image(np.zeros((4, 4), np.uint8))
    assert L.lib().tb_resample_ksize(0, 4, D.LANCZOS) == -22 and L.lib().tb_resample_ksize(8, 4, 2) == -22
    with pytest.raises(RuntimeError):
        ops.img_to_pixels(t, 10, 10, torch.empty(3, 8, 8, device="cuda"))  # window leaves the image
    with pytest.raises(RuntimeError):
        ops.img_affine_bicubic(t, [1.0, 0.5, 0.0, 0.0, 1.0, 0.0], 0, 0, 0, 0, 16, 16)  # rotation / shear is not on this path


def test_cli_trains_from_image_files_through_the_device_feeder(tmp_path, monkeypatch):
    """train_textboost.py on the reference's own inputs -- image files + a tokenizer: images decoded once, the per-sample dataset work
    (template draw, PairedAugmentation, Lanczos resize, crop, normalise, tokenise) on the device, feeding the device VAE encoder."""
    import sys
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    from tests.test_host_logic import _WordTokenizer
    data = tmp_path / "dog"
    data.mkdir()
    r = np.random.default_rng(0)
    for i, (h, w) in enumerate([(300, 260), (200, 340)]):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.clip(np.stack([xx * 255 // w, yy * 255 // h, (xx + yy) % 256], -1) + r.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(a).save(str(data / f"{i:02d}.png"))
    tok = _WordTokenizer()
    monkeypatch.setattr(T, "load_tokenizer", lambda mdir, name=None: tok)
    # the knowledge-preservation prompts come from the reference's hard-coded relative path (train_textboost.py:892)
    (tmp_path / "data").mkdir()
    (tmp_path / "data" / "human-written-prompts.jsonl").write_text(
        "\n".join(json.dumps({"input": f"make the {w} blue", "output": f"a blue {w}"}) for w in ("car", "house", "sky", "boat")) + "\n")
    monkeypatch.chdir(tmp_path)
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--instance_data_dir", str(data), "--output_dir", out,
                         "--train_batch_size", "2", "--resolution", "128", "--max_train_steps", "4", "--placeholder_token", "<dog>",
                         "--initializer_token", "dog", "--lora_rank", "4", "--mixed_precision", "fp16", "--seed", "3", "--augment", "paug",
                         "--augment_inversion", "--augment_p", "0.9", "--template", "textboost"])
    T.main(args)
    log = open(os.path.join(out, "training.log")).read()
    assert "device feeder: 2 resident instance image(s)" in log and "VAE" in log
    assert "prior prompts: 8 edit prompts + 5 template prompts" in log
    d = torch.load(os.path.join(out, "dog.bin"))
    assert torch.isfinite(d["<dog>"]).all()
    # the augmentation tokens were registered through the tokenizer (word-level stand-in: 11 vectors) and saved next to the placeholder
    assert os.path.exists(os.path.join(out, "hflip.bin")) and os.path.exists(os.path.join(out, "zoom-in_0.bin"))
    assert "<dog>" in tok.vocab and "<left>" in tok.vocab


def test_resample_24bit_and_32bit_multiply_paths_agree():
    """The full-rate 24-bit multiply path is taken only when the host table fits (it always does for real images); the 32-bit path is the
    general one.  Both must give Pillow's bytes."""
    from oracle import augment as A
    from textboost_amd import augment as D
    from textboost_amd import ops
    a = rnd(77, 120, 333)
    t = up(a)
    for filt_o, filt_d, n_out in ((A.LANCZOS, D.LANCZOS, 100), (A.BICUBIC, D.BICUBIC, 500)):
        _, b, kk = ops.resample_coeffs(333, n_out, filt_d)
        assert ops.resample_fit24(kk)
        want = A._resample_axis(a, n_out, filt_o, 1)
        for fit in (False, True):
            assert np.array_equal(down(ops.img_resample(t, n_out, b.cuda(), kk.cuda(), 0, fit)), want)
        _, b, kk = ops.resample_coeffs(120, n_out // 4, filt_d)
        want = A._resample_axis(a, n_out // 4, filt_o, 0)
        for fit in (False, True):
            assert np.array_equal(down(ops.img_resample(t, n_out // 4, b.cuda(), kk.cuda(), 1, fit)), want)
    assert not ops.resample_fit24(torch.tensor([[1 << 23]], dtype=torch.int32))


def test_random_shape_sweep_of_the_whole_item_pipeline():
    """24 seeded random image shapes (portrait / landscape / tiny / odd sizes, up- and down-scaling) through augmentation + Lanczos resize + random
    crop + normalise: device vs oracle, bit-exact pixel_values and equal prompts."""
    from oracle import augment as A
    from textboost_amd import augment as D
    r = np.random.default_rng(2024)
    templates = ["a {}", "photo of a {}"]
    for case in range(24):
        h, w = int(r.integers(20, 400)), int(r.integers(20, 400))
        size = int(r.choice([16, 24, 33, 64]))
        a = rnd(5000 + case, h, w)
        cfg = dict(hflip=str(r.choice(["false", "true", "inversion"])), inversion=bool(r.integers(0, 2)), p=0.9, color_prob=0.5,
                   ops=str(r.choice(["object", "style"])))
        center = bool(r.integers(0, 2))
        feeder = D.DeviceFeeder([(up(a), "<x>")], FakeTokenizer(), templates, size=size, center_crop=center, augment_pipe=D.PairedAugmentation(**cfg))
        random.seed(case), np.random.seed(case), torch.manual_seed(case)
        got = feeder.batch([0, 0])
        random.seed(case), np.random.seed(case), torch.manual_seed(case)
        pipe = A.PairedAugmentation(**cfg)
        for b in range(2):
            pv, prompt = A.dataset_item(a, "<x>", templates, size, center, pipe)
            assert got["prompts"][b] == prompt, (case, b)
            assert np.array_equal(got["pixel_values"][b].cpu().numpy(), pv), (case, b, h, w, size, cfg, center)


@pytest.mark.parametrize("G", [1, 2])
def test_cli_resume_replays_the_feeder_streams(tmp_path, monkeypatch, G):
    """Resume (train_textboost.py:959-981): a run interrupted at checkpoint-2 and resumed to step 4 sees the same samples (index streams
    fast-forwarded, torch / numpy / `random` states restored from random_states_0.pkl) as the uninterrupted run -- also with
    --gradient_accumulation_steps G > 1, where an optimizer step consumed G batches."""
    import sys
    pytest.importorskip("PIL")
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    from tests.test_host_logic import _WordTokenizer
    data = tmp_path / "dog"
    data.mkdir()
    for i, (h, w) in enumerate([(160, 140), (150, 190), (200, 200)]):
        Image.fromarray(rnd(40 + i, h, w)).save(str(data / f"{i}.png"))
    monkeypatch.chdir(tmp_path)
    base = ["--pretrained_model_name_or_path", "/nonexistent/sd15", "--instance_data_dir", str(data), "--train_batch_size", "2", "--resolution",
            "64", "--placeholder_token", "<dog>", "--initializer_token", "dog", "--lora_rank", "4", "--mixed_precision", "fp16", "--seed", "11",
            "--augment", "paug", "--augment_inversion", "--augment_p", "0.9", "--checkpointing_steps", "2", "--emb_learning_rate", "1e-2",
            "--gradient_accumulation_steps", str(G)]
    prompts = {}

    def run(out, extra):
        monkeypatch.setattr(T, "load_tokenizer", lambda mdir, name=None: _WordTokenizer())
        seen = []
        from textboost_amd import augment as D
        orig = D.DeviceFeeder.batch

        def spy(self, indices, out=None):
            b = orig(self, indices, out=out)
            seen.append((list(indices), list(b["prompts"])))
            return b
        monkeypatch.setattr(D.DeviceFeeder, "batch", spy)
        T.main(T.parse_args(base + ["--output_dir", out] + extra))
        monkeypatch.setattr(D.DeviceFeeder, "batch", orig)
        prompts.setdefault(out, []).extend(seen)
        return torch.load(os.path.join(out, "dog.bin"))["<dog>"]

    full = run(str(tmp_path / "full"), ["--max_train_steps", "4"])
    run(str(tmp_path / "part"), ["--max_train_steps", "2"])
    resumed = run(str(tmp_path / "part"), ["--max_train_steps", "4", "--resume_from_checkpoint", "latest"])
    a, bc = prompts[str(tmp_path / "full")], prompts[str(tmp_path / "part")]
    assert len(a) == 4 * G and bc == a  # same indices, same augmented prompts, batch by batch across the interruption
    assert torch.allclose(full, resumed, rtol=0, atol=2e-3) and not torch.equal(full, torch.zeros_like(full))


def test_prefetch_feeder_matches_direct_batches():
    """PrefetchFeeder (side stream + staging) delivers exactly what DeviceFeeder.batch would have, in order."""
    from textboost_amd import augment as D
    imgs = [rnd(61, 130, 170), rnd(62, 200, 120)]
    templates = ["a {}", "{} on a table"]

    def make():
        return D.DeviceFeeder([(up(a), "<s>") for a in imgs], FakeTokenizer(), templates, size=48, center_crop=False,
                              augment_pipe=D.PairedAugmentation(hflip="inversion", inversion=True, p=0.9, color_prob=0.5))
    idx = [[0, 1, 1], [1, 0, 0], [0, 0, 1], [1, 1, 0]]
    random.seed(8), np.random.seed(8), torch.manual_seed(8)
    f = make()
    want = [f.batch(i) for i in idx]
    want = [(b["pixel_values"].clone(), b["input_ids"].clone(), b["prompts"]) for b in want]
    random.seed(8), np.random.seed(8), torch.manual_seed(8)
    pv = torch.zeros(3, 3, 48, 48, device="cuda")
    ids = torch.zeros(3, 77, dtype=torch.int64, device="cuda")
    pre = D.PrefetchFeeder(make(), 3, pv, ids)
    pre.prefetch(idx[0])
    for k in range(4):
        prompts = pre.commit()
        if k + 1 < 4:
            pre.prefetch(idx[k + 1])  # overlaps with the consumer below, as in the training loop
        busy = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")  # stand-in for the step on the main stream
        assert prompts == want[k][2]
        assert torch.equal(pv, want[k][0]) and torch.equal(ids.cpu(), want[k][1])
        del busy
    with pytest.raises(RuntimeError):
        pre.commit()


def test_cli_concepts_list_trains_two_concepts(tmp_path, monkeypatch):
    """--concepts_list (train_textboost.py:602-615, :661-694; dataset.py:302-308): two concepts, each with its own image directory and
    placeholder token -- every concept's images carry that concept's token list, both placeholder rows are registered, trained and saved;
    --tokenizer_name is accepted (:630-633)."""
    import sys
    pytest.importorskip("PIL")
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import train_textboost as T
    from tests.test_host_logic import _WordTokenizer
    r = np.random.default_rng(1)
    concepts = []
    for name, init in (("<dog>", "dog"), ("<cat>", "cat")):
        d = tmp_path / name.strip("<>")
        d.mkdir()
        a = r.integers(0, 256, (150, 170, 3)).astype(np.uint8)
        Image.fromarray(a).save(str(d / "00.png"))
        concepts.append({"instance_data_dir": str(d), "placeholder_token": name, "initializer_token": init})
    cl = tmp_path / "concepts.json"
    cl.write_text(json.dumps(concepts))
    tok = _WordTokenizer()
    seen = {}
    monkeypatch.setattr(T, "load_tokenizer", lambda mdir, name=None: seen.setdefault("name", name) and tok or tok)
    monkeypatch.chdir(tmp_path)
    out = str(tmp_path / "run")
    args = T.parse_args(["--pretrained_model_name_or_path", "/nonexistent/sd15", "--concepts_list", str(cl), "--output_dir", out,
                         "--train_batch_size", "2", "--resolution", "128", "--max_train_steps", "3", "--lora_rank", "4",
                         "--mixed_precision", "fp16", "--seed", "3", "--template", "textboost", "--tokenizer_name", str(tmp_path / "tok")])
    T.main(args)
    assert seen["name"] == str(tmp_path / "tok")
    log = open(os.path.join(out, "training.log")).read()
    assert "device feeder: 2 resident instance image(s)" in log
    for n in ("dog", "cat"):
        v = torch.load(os.path.join(out, n + ".bin"))
        assert torch.isfinite(v[f"<{n}>"]).all()
    assert "<dog>" in tok.vocab and "<cat>" in tok.vocab
