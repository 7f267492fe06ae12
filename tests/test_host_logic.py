"""CPU-only checks of the host side: shape specs, C-ABI export table, loader behaviour (no GPU compute)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shape_specs_match_published_param_counts_and_oracle_keys():
    from textboost_amd import models
    from oracle.unet_sd import UNet2DCondition, UNetConfig
    from oracle.clip_text import CLIPTextCfg, TextBoostEncoder
    s = models.unet_shapes(models.SD15_UNET)
    assert models.count_params(s) == 859_520_964
    with torch.device("meta"):
        ref = UNet2DCondition(UNetConfig.sd15())
    rs = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert rs == s
    s21 = models.unet_shapes(models.SD21_UNET)
    with torch.device("meta"):
        ref = UNet2DCondition(UNetConfig.sd21())
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == s21
    c = models.clip_shapes(models.SD15_CLIP)
    assert models.count_params(c) == 123_060_480
    with torch.device("meta"):
        enc = TextBoostEncoder(CLIPTextCfg.sd15())
    assert set(enc.hf_key_map().values()) == set(c.keys())


def test_library_exports_every_symbol_in_header():
    from textboost_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from textboost_amd.build import build
        build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "textboost_hip.h")).read()
    names = set(re.findall(r"\b(tb_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    for path in (_lib.LIB_PATH, _lib.LIB_PATH_BF16):   # the fp16 build and the bfloat16 build (-DTB_BF16) share one C-ABI
        lib = ctypes.CDLL(path)
        for n in sorted(names):
            assert hasattr(lib, n), f"{n} declared in include/textboost_hip.h but not exported by {os.path.basename(path)}"
    assert names == set(_lib._SIGS.keys()), names ^ set(_lib._SIGS.keys())


def test_half_kind_switch_selects_the_library_and_the_dtype():
    from textboost_amd import _lib
    assert _lib.half_kind() == "fp16" and _lib.half_dtype() == torch.float16
    prev = _lib.set_half("bf16")
    try:
        assert prev == "fp16" and _lib.half_dtype() == torch.bfloat16
        h = _lib.lib()
        assert h._name == _lib.LIB_PATH_BF16
    finally:
        _lib.set_half("fp16")
    assert _lib.lib()._name == _lib.LIB_PATH and _lib.half_dtype() == torch.float16


def test_missing_library_fails_loudly(monkeypatch):
    from textboost_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtextboost_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_code_never_imports_oracle():
    pkg = os.path.join(ROOT, "textboost_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("ORACLE", ""), f


def test_lr_lambda_matches_transformers_schedules():
    """diffusers.optimization.get_scheduler (train_textboost.py:911-916) is the schedule family of transformers.optimization: pin the
    host-side multipliers against the installed transformers implementation for every name the reference's --lr_scheduler accepts."""
    import torch
    from transformers.optimization import get_scheduler
    from textboost_amd.trainer import lr_lambda
    W, T, lr0 = 7, 40, 5e-5
    for name in ["constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial"]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr0)
        sch = get_scheduler(name, optimizer=opt, num_warmup_steps=W, num_training_steps=T)
        lam = lr_lambda(name, W, T, lr_init=lr0)
        for step in range(T + 3):
            assert abs(sch.get_last_lr()[0] - lr0 * lam(step)) <= 1e-12 + 1e-9 * lr0, (name, step, sch.get_last_lr()[0], lr0 * lam(step))
            opt.step(); sch.step()


# ------------------------------------------------------------------------------ instance-image data path host logic
class _WordTokenizer:
    """Duck-typed stand-in for CLIPTokenizer (no tokenizer files exist offline): one id per whitespace word, grows with add_tokens."""
    model_max_length = 77

    def __init__(self):
        self.vocab = {}
        self.base = 49408

    def _id(self, w):
        if w in self.vocab:
            return self.vocab[w]
        return sum(map(ord, w)) % 49000

    def encode(self, text, add_special_tokens=False):
        return [self._id(w) for w in text.split()]

    def add_tokens(self, names):
        n = 0
        for t in names:
            if t not in self.vocab:
                self.vocab[t] = self.base + len(self.vocab)
                n += 1
        return n

    def convert_tokens_to_ids(self, names):
        return [self.vocab[t] for t in names]

    def __call__(self, prompt, truncation=True, padding="max_length", max_length=77, return_tensors="pt"):
        import types
        ids = ([49406] + self.encode(prompt))[:max_length - 1]
        ids = ids + [49407] * (max_length - len(ids))
        return types.SimpleNamespace(input_ids=torch.tensor([ids], dtype=torch.int64))


class _Table:
    """the part of HipTextEncoder.add_tokens the registration uses"""

    def __init__(self, V=49408):
        self.V, self.copied = V, []

    def add_tokens(self, init_ids):
        new = list(range(self.V, self.V + len(init_ids)))
        self.V += len(init_ids)
        self.copied += list(init_ids)
        return new


def test_index_stream_is_the_reference_wrapper_order(golden_dir):
    from textboost_amd.data import IndexStream
    g = torch.load(os.path.join(golden_dir, "wrapper_order.pt"))
    assert IndexStream(5, 42).take(10) == g["n5_seed42_rep2"]
    assert IndexStream(1, 42).take(4) == g["n1_seed42_rep4"]
    # two ranks, five samples: padded with the head of the epoch, strided; one image: every rank still gets it (drop_last=False)
    a, b = IndexStream(5, 42, 0, 2), IndexStream(5, 42, 1, 2)
    e0 = [4, 2, 3, 1, 0]
    assert a.take(3) == (e0 + e0[:1])[0::2] and b.take(3) == (e0 + e0[:1])[1::2]
    assert IndexStream(1, 0, 1, 2).take(3) == [0, 0, 0]


def test_templates_token_registration_and_image_listing(tmp_path):
    from textboost_amd import data as D
    assert D.load_templates("textboost") == ["{}", "a {}", "one {}", "the {}", "photo of a {}"]
    assert len(D.load_templates("imagenet_small")) == 27 and len(D.load_templates("imagenet_style_small")) == 19
    assert D.load_templates("a photo of {} dog") == ["a photo of {} dog"]
    assert D.multi_vector_names("<dog>", 1) == ["<dog>"] and D.multi_vector_names("<dog>", 3) == ["<dog_0>", "<dog_1>", "<dog_2>"]
    assert D.multi_vector_names("sks", 2) == ["sks", "sks_1"]
    tok, table = _WordTokenizer(), _Table()
    names, ids = D.add_token(table, tok, "<sks>", "dog")
    assert names == ["<sks>"] and ids == [49408] and table.copied == tok.encode("dog")
    aug_ids, aug = D.add_augmentation_tokens(table, tok, "object")
    assert list(aug) == ["<grayscale>", "<zoom-in_0>", "<zoom-in_1>", "<zoom-out_0>", "<zoom-out_1>", "<collage_0>", "<collage_1>", "<crop>",
                         "<hflip>", "<left>", "<right>"]  # word-level stand-in: 2 pieces for the two-word initialisers
    assert aug_ids == list(range(49409, 49409 + 11)) and table.V == 49408 + 12
    with pytest.raises(ValueError):
        D.add_token(table, tok, "<sks>", "dog")  # already registered (utils.py:145-149)
    (tmp_path / "b.png").write_bytes(b"x")
    (tmp_path / "a.jpg").write_bytes(b"x")
    (tmp_path / "c.jpg").write_bytes(b"x")
    assert [os.path.basename(p) for p in D.get_images_path(str(tmp_path), 2)] == ["a.jpg", "b.png"]
    assert D.has_instance_images(str(tmp_path)) and not D.has_instance_images(str(tmp_path / "nope"))
    with pytest.raises(ValueError):
        D.get_images_path(str(tmp_path / "nope"))


def test_prior_prompt_feeder_follows_prior_dataset(tmp_path):
    """PriorDataset.__getitem__ / collate_fn (dataset.py:196-269) and InstructPix2PixDataset's file format (:162-175)."""
    import json as js
    import random
    from textboost_amd import data as D
    from textboost_amd.augment import PromptFeeder
    f = tmp_path / "p.jsonl"
    f.write_text("\n".join(js.dumps(r) for r in [{"input": "a cat", "output": "a dog"}, {"input": "make it red", "output": "NONE"},
                                                   {"input": "add snow", "output": None}, {"input": "x y", "output": "z"}]) + "\n")
    prompts = D.read_edit_prompts(str(f))
    assert prompts == ["a cat", "a dog", "make it red", "add snow", "x y", "z"]
    assert D.read_edit_prompts(str(f), 3) == ["a cat", "a dog", "make it red"]
    tok = _WordTokenizer()
    feeder = D.PriorPromptFeeder(prompts, PromptFeeder(tok), additional_template="textboost", additional_category=["dog", "puppy"],
                                 template_prob=0.3, null_prob=0.2, seed=42)
    assert feeder.template_data == ["dog", "puppy", "a dog", "a puppy", "one dog", "one puppy", "the dog", "the puppy", "photo of a dog",
                                    "photo of a puppy"]
    random.seed(9)
    b = feeder.batch(40)
    # replay the reference's logic with the same draws and the same Wrapper order
    random.seed(9)
    order = D._DropLastIndexStream(len(prompts), 42).take(40)
    want = []
    for i in order:
        r = random.random()
        want.append("" if r < 0.2 else random.choice(feeder.template_data) if r < 0.5 else prompts[i])
    assert b["prompt"] == want and "" in want and any(w in feeder.template_data for w in want)
    assert b["input_ids"].shape == (40, 77) and b["input_ids"].dtype == torch.int64
    assert torch.equal(b["input_ids"][want.index("")], torch.tensor([49406] + [49407] * 76))  # the null prompt
    # drop_last: 6 prompts over 4 ranks -> each epoch 4 of them, one per rank
    got = [D._DropLastIndexStream(6, 1, r, 4).take(1)[0] for r in range(4)]
    assert len(set(got)) == 4
    with pytest.raises(ValueError):
        D.PriorPromptFeeder(["a"], PromptFeeder(tok), additional_template="{}", additional_category=None, world=2)


def test_geometry_from_published_model_configs():
    """unet/config.json and text_encoder/config.json of SD1.5 and SD2.1 (the fields diffusers / transformers publish) -> the built geometries."""
    from textboost_amd import models
    sd15_unet = {"act_fn": "silu", "attention_head_dim": 8, "block_out_channels": [320, 640, 1280, 1280], "center_input_sample": False,
                 "cross_attention_dim": 768, "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "downsample_padding": 1,
                 "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2, "mid_block_scale_factor": 1,
                 "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 64,
                 "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3}
    sd21_unet = dict(sd15_unet, attention_head_dim=[5, 10, 20, 20], cross_attention_dim=1024, use_linear_projection=True, sample_size=96,
                     dual_cross_attention=False, only_cross_attention=False, upcast_attention=False)
    assert models.unet_geometry_from_config(sd15_unet) == models.SD15_UNET
    assert models.unet_geometry_from_config(sd21_unet) == models.SD21_UNET
    clip_l = {"hidden_act": "quick_gelu", "hidden_size": 768, "intermediate_size": 3072, "layer_norm_eps": 1e-05, "max_position_embeddings": 77,
              "num_attention_heads": 12, "num_hidden_layers": 12, "vocab_size": 49408}
    clip_h = dict(clip_l, hidden_act="gelu", hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=23)
    assert models.clip_geometry_from_config(clip_l) == models.SD15_CLIP
    assert models.clip_geometry_from_config(clip_h) == models.SD21_CLIP
    with pytest.raises(NotImplementedError):
        models.unet_geometry_from_config(dict(sd15_unet, addition_embed_type="text_time"))       # SDXL
    with pytest.raises(NotImplementedError):
        models.unet_geometry_from_config(dict(sd15_unet, down_block_types=["DownBlock2D"] * 4))  # mirror check: up blocks still cross-attn
    with pytest.raises(NotImplementedError):
        models.clip_geometry_from_config(dict(clip_l, hidden_act="relu"))


# ------------------------------------------------------------------------------ bench.py --gpus N launch logic (run_textboost_db.py:106-111)
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_gpus_flag_decides_the_launch():
    """`python bench.py --gpus N` with no launcher around it must itself start N ranks (the reference's driver runs
    `torchrun --nproc-per-node=len(gpus)`), must refuse N > visible GPUs, and inside an existing launch must agree with WORLD_SIZE."""
    b = _bench()
    assert b.plan_launch(1, {}, 1) == {"action": "run", "world": 1, "rank": 0, "local": 0}
    assert b.plan_launch(8, {}, 8) == {"action": "spawn", "world": 8}
    p = b.plan_launch(2, {}, 1)  # a 1-GPU box asked for 2 GPUs: loud error, never a silent 1-GPU run
    assert p["action"] == "error" and "exposes 1 GPU" in p["message"]
    env = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}
    assert b.plan_launch(4, env, 8) == {"action": "run", "world": 4, "rank": 3, "local": 3}
    assert b.plan_launch(8, env, 8)["action"] == "error"          # --gpus disagrees with the launcher
    assert b.plan_launch(4, env, 2)["action"] == "error"          # LOCAL_RANK 3 on a 2-GPU node
    assert b.plan_launch(0, {}, 8)["action"] == "error"
    cmd = b.launcher_command(8, ["--gpus", "8", "--steps", "5"], 29999, python="py", script="bench.py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                   "--master-port", "29999", "bench.py", "--gpus", "8", "--steps", "5"]
    assert 1024 < b.free_port() < 65536


def test_bench_spawn_starts_n_ranks_with_the_rank_environment(tmp_path, monkeypatch):
    """spawn_ranks() end to end on CPU: the launcher command it builds really starts N processes that see RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 (a stand-in script takes bench.py's place: no GPU here)."""
    import subprocess
    b = _bench()
    script = tmp_path / "probe.py"
    script.write_text("import os,sys\nopen(os.path.join(sys.argv[1], 'r%s' % os.environ['RANK']), 'w').write("
                      "'%s %s %s %s' % (os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'], os.environ['MASTER_ADDR'], sys.argv[2]))\n")
    cmd = b.launcher_command(2, [str(tmp_path), "--gpus=2"], b.free_port(), script=str(script))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "r0").read_text() == "2 0 127.0.0.1 --gpus=2"
    assert (tmp_path / "r1").read_text() == "2 1 127.0.0.1 --gpus=2"


def test_layernorm_epilogue_query_follows_the_row_spanning_tiles():
    """tb_gemm_ln_epilogue_ok (host side of the library, no GPU needed): the fused LayerNorm epilogues exist exactly where a wide tile spans the
    whole output row -- N = 320 with at least one chip round of 128- or 64-row tiles (the 64x64 maps at B >= 4 / 2); everything else keeps the
    separate tb_layernorm_* launches."""
    from textboost_amd import _lib as L
    ok = lambda M, N, K: bool(L.lib().tb_gemm_ln_epilogue_ok(M, N, K))  # noqa: E731
    assert ok(8 * 4096, 320, 320) and ok(8 * 4096, 320, 2560) and ok(16 * 4096, 320, 960)       # metric batch, B = 16
    assert ok(4 * 4096, 320, 320) and ok(200 * 64, 320, 64)                                      # 64-row tiles from 200 tiles on
    assert not ok(4096, 320, 320) and not ok(199 * 64, 320, 320)                                 # less than one round: B = 1
    assert not ok(8 * 1024, 640, 640) and not ok(8 * 4096, 160, 320)                             # no tile spans N = 640; N must be 320
    assert not ok(8 * 4096, 320, 100) and not ok(8 * 4096 + 8, 320, 320)                         # K % 64, ragged M


def test_bench_classifies_every_recorded_kernel_family():
    """`roofline.classes` (bench.py): every launch record of the eager leg falls into conv / linear / attention / norm by its rocprofv3 symbol."""
    b = _bench()
    want = {"gemm8_kernel<4, 2, 4, 5, true, 3>": "conv", "gemm8_kernel<8, 1, 2, 5, true, 4>": "conv", "conv_halo_kernel<128>": "conv",
            "gemm_kernel<128, 128, 1, 64, 2>": "conv", "gemm8_kernel<4, 2, 2, 5, false, 4>": "linear", "gemm_kernel<64, 64, 0, 64, 4>": "linear",
            "lin320_kernel<true>": "linear", "ff_fused_kernel<false>": "linear", "gemm_f32_kernel": "linear", "attn_fwd_kernel": "attention",
            "attn_bwd(delta+dq+dkv)": "attention", "groupnorm_fwd(splitk)": "norm", "groupnorm_bwd(stats+apply)": "norm",
            "layernorm_fwd": "norm", "layernorm_bwd": "norm"}
    for name, cls in want.items():
        assert b.kernel_class(name) == cls, name
