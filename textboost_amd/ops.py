"""Thin Python launch wrappers over the C-ABI kernels (one function per entry point of include/textboost_hip.h).
Tensors are PyTorch-ROCm tensors used for memory + streams only; all arithmetic happens in the HIP kernels."""
from __future__ import annotations

import torch

from . import _lib as L


def _dt(t):
    return L.TB_F32 if t.dtype == torch.float32 else L.TB_F16


def gemm(A, W, out, *, K=None, A2=None, W2=None, bias=None, rowbias=None, rows_per_group=0, R=None, act=L.ACT_NONE,
         alpha=1.0, C2=None, conv=None):
    """out[M,N] = A[M,K] @ W[N,K]^T (+epilogue). A/out/R may be column-slices of wider buffers (stride(0) = ld).
    conv = dict(B,Hin,Win,Cin,Hout,Wout,stride,sign,upsample,transposed) switches A to an NHWC 3x3 gather."""
    d = L.GemmDesc()
    M, Nout = out.shape
    N = W.shape[0]
    d.M, d.N = M, N
    d.A, d.lda = L.ptr(A), A.stride(0)
    d.W, d.ldw = L.ptr(W), W.stride(0)
    if conv is not None:
        d.a_mode = L.A_CONV3X3
        for k, v in conv.items():
            setattr(d, k, v)
        d.K = 9 * conv["Cin"]
        d.K1 = d.K
        d.lda = A.stride(-2)
    else:
        d.a_mode = L.A_LINEAR
        K1 = W.shape[1]
        d.K1 = K1
        d.K = K1 + (W2.shape[1] if W2 is not None else 0)
        if A2 is not None:
            d.A2, d.lda2 = L.ptr(A2), A2.stride(0)
            d.W2, d.ldw2 = L.ptr(W2), W2.stride(0)
    d.alpha = alpha
    d.bias = L.ptr(bias)
    d.rowbias = L.ptr(rowbias)
    d.rows_per_group = rows_per_group
    if R is not None:
        d.R, d.ldr, d.r_dtype = L.ptr(R), R.stride(0), _dt(R)
    d.act = act
    d.C, d.ldc, d.c_dtype = L.ptr(out), out.stride(0), _dt(out)
    if C2 is not None:
        d.C2, d.ldc2 = L.ptr(C2), C2.stride(0)
    L.check(L.lib().tb_gemm(d, L.stream()), "tb_gemm")
    return out


def groupnorm_ws(B, HW, C, G=32):
    return int(L.lib().tb_groupnorm_ws_floats(B, HW, C, G))


def groupnorm_fwd(x, y, gamma, beta, stats, ws, B, HW, C, G=32, eps=1e-5, silu=False):
    """x,y: [B*HW, C] fp16 (may be strided slices). stats [B,G,2] fp32 out."""
    L.check(L.lib().tb_groupnorm_fwd(L.ptr(x), x.stride(0), L.ptr(y), y.stride(0), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                     L.ptr(ws), B, HW, C, G, eps, int(silu), L.stream()), "tb_groupnorm_fwd")
    return y


def groupnorm_bwd(dy, x, gamma, beta, stats, dx, ws, B, HW, C, G=32, silu=False, add=None):
    L.check(L.lib().tb_groupnorm_bwd(L.ptr(dy), dy.stride(0), L.ptr(x), x.stride(0), L.ptr(gamma), L.ptr(beta), L.ptr(stats),
                                     L.ptr(add), add.stride(0) if add is not None else 0, L.ptr(dx), dx.stride(0), L.ptr(ws),
                                     B, HW, C, G, int(silu), L.stream()), "tb_groupnorm_bwd")
    return dx


def layernorm_fwd(x, y, gamma, beta, stats, eps=1e-5):
    M, Cc = y.shape
    L.check(L.lib().tb_layernorm_fwd(L.ptr(x), x.stride(0), _dt(x), L.ptr(y), y.stride(0), L.ptr(gamma), L.ptr(beta),
                                     L.ptr(stats), M, Cc, eps, L.stream()), "tb_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, gamma, stats, dx, add=None):
    M, Cc = dx.shape
    L.check(L.lib().tb_layernorm_bwd(L.ptr(dy), dy.stride(0), _dt(dy), L.ptr(x), x.stride(0), _dt(x), L.ptr(gamma), L.ptr(stats),
                                     L.ptr(add), add.stride(0) if add is not None else 0, L.ptr(dx), dx.stride(0), M, Cc,
                                     L.stream()), "tb_layernorm_bwd")
    return dx


def _attn_desc(q, k, v, o, lse, B, H, Sq, Skv, hd, scale, causal):
    d = L.AttnDesc()
    d.B, d.H, d.Sq, d.Skv, d.hd, d.causal, d.scale = B, H, Sq, Skv, hd, int(causal), scale
    d.Q, d.ldq = L.ptr(q), q.stride(0)
    d.K, d.ldk = L.ptr(k), k.stride(0)
    d.V, d.ldv = L.ptr(v), v.stride(0)
    d.O, d.ldo = L.ptr(o), o.stride(0)
    d.LSE = L.ptr(lse)
    return d


def attention_fwd(q, k, v, o, lse, B, H, Sq, Skv, hd, scale=None, causal=False):
    """q,o: [B*Sq, H*hd]; k,v: [B*Skv, H*hd] fp16 (column slices allowed). lse fp32 [B,H,Sq]."""
    d = _attn_desc(q, k, v, o, lse, B, H, Sq, Skv, hd, scale if scale is not None else hd ** -0.5, causal)
    L.check(L.lib().tb_attention_fwd(d, L.stream()), "tb_attention_fwd")
    return o


def attention_bwd(q, k, v, o, lse, do, delta, dq, dk, dv, B, H, Sq, Skv, hd, scale=None, causal=False):
    d = _attn_desc(q, k, v, o, lse, B, H, Sq, Skv, hd, scale if scale is not None else hd ** -0.5, causal)
    d.dO, d.lddo = L.ptr(do), do.stride(0)
    d.Delta = L.ptr(delta)
    d.dQ, d.lddq = L.ptr(dq), dq.stride(0)
    d.dK, d.lddk = L.ptr(dk), dk.stride(0)
    d.dV, d.lddv = L.ptr(dv), dv.stride(0)
    L.check(L.lib().tb_attention_bwd(d, L.stream()), "tb_attention_bwd")
