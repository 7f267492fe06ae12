"""Parity metrics shared by the GPU tests: whole-tensor rel-L2 AND max-abs (relative to the reference's largest magnitude) AND the worst
per-channel rel-L2 -- a whole-tensor rel-L2 alone hides a wrong channel, a dropped tap on a border row or a mis-scaled head."""
import torch


def parity(name, got, ref, rel, maxabs=None, ch_dim=None, ch_rel=None, verbose=True):
    a, b = got.detach().float().cpu(), ref.detach().float().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    r = ((a - b).norm() / (b.norm() + 1e-30)).item()
    m = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    msg = f"{name}: rel-L2 {r:.3e} (<= {rel:.1e})  max-abs/max|ref| {m:.3e}" + (f" (<= {maxabs:.1e})" if maxabs else "")
    c = None
    if ch_dim is not None:
        dims = [d for d in range(a.dim()) if d != ch_dim]
        c = ((a - b).pow(2).sum(dims).sqrt() / b.pow(2).sum(dims).sqrt().clamp_min(1e-30)).max().item()
        msg += f"  worst-channel rel-L2 {c:.3e}" + (f" (<= {ch_rel:.1e})" if ch_rel else "")
    if verbose:
        print("[parity]", msg)
    assert r <= rel, msg
    if maxabs is not None:
        assert m <= maxabs, msg
    if ch_rel is not None:
        assert c <= ch_rel, msg
    return r, m, c
