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
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/textboost_hip.h but not exported"
    assert names == set(_lib._SIGS.keys()), names ^ set(_lib._SIGS.keys())


def test_missing_library_fails_loudly(monkeypatch):
    from textboost_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
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
