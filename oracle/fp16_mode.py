"""ORACLE (test infrastructure only) -- fp16-faithful evaluation of the fp32 restatements.

The reference runs the UNet and the KPL teacher as plain fp16 modules (`unet.to(weight_dtype)`, train_textboost.py:937-939): every
operator reads fp16 tensors, accumulates in fp32 inside the kernel and ROUNDS ITS RESULT TO fp16.  `fp16_rounding()` reproduces exactly
that on the CPU without fp16 kernels: a TorchDispatchMode that runs every ATen operator in fp32 and rounds each floating-point result to
the nearest fp16 value (forward and backward alike).  With it the oracle carries the same rounding points as the reference's fp16 run,
so the GPU parity tests no longer have to fold "fp16 vs fp32" into their tolerance.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_map

_SKIP = ("aten.detach", "aten.view", "aten._unsafe_view", "aten.reshape", "aten.t.", "aten.transpose", "aten.permute", "aten.expand",
         "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.as_strided", "aten.split", "aten.chunk",
         "aten.unbind", "aten.clone", "aten.contiguous", "aten._to_copy", "aten.copy_", "aten.zeros", "aten.ones", "aten.empty",
         "aten.full", "aten.arange", "aten.cat", "aten.stack", "aten.index", "aten.embedding.", "aten.lift_fresh")


class fp16_rounding(TorchDispatchMode):
    """`with fp16_rounding(): y = module(x)`: every operator result (fp32) is rounded to fp16 precision.
    `keep_fp32(fn)`-style escape: tensors with `requires_fp32 = True` attribute are left alone (not used by default)."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(_SKIP):
            return out

        def rnd(t):
            if isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.numel() > 0:
                return t.to(torch.float16).to(torch.float32)
            return t
        return tree_map(rnd, out)
