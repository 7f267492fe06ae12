"""ctypes binding of libtextboost_hip.so (see include/textboost_hip.h). Fails loudly if the library is missing:
there is no CPU / eager fallback in the product path."""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first so libamdhip64.so.7 resolves to the runtime torch uses)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libtextboost_hip" + os.environ.get("TB_LIB_SUFFIX", "") + ".so")
# the same sources compiled with -DTB_BF16: every 16-bit operand / activation is bfloat16 (the reference's --mixed_precision bf16)
LIB_PATH_BF16 = os.path.join(_PKG, "libtextboost_hip_bf16" + os.environ.get("TB_LIB_SUFFIX", "") + ".so")

TB_F16, TB_F32 = 0, 1
ACT_NONE, ACT_QUICK_GELU, ACT_GEGLU, ACT_SILU, ACT_QUICK_GELU_GRAD, ACT_GELU, ACT_GELU_GRAD, ACT_GEGLU_GRAD = 0, 1, 2, 3, 4, 5, 6, 7
ACT_LN_FWD, ACT_LN_BWD = 8, 9
(ST_LOSS_SCALE, ST_GROWTH_TRACKER, ST_STEP, ST_FOUND_INF, ST_COEF_LORA, ST_COEF_EMB, ST_BC1, ST_BC2, ST_GRAD_NORM,
 ST_SUMSQ_LORA, ST_SUMSQ_EMB, ST_LOSS_MSE, ST_LOSS_KPL, ST_LR_MULT) = range(14)
ST_COUNT = 16
A_LINEAR, A_CONV3X3 = 0, 1


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("A2", C.c_void_p), ("lda2", C.c_int64),
        ("K1", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("W2", C.c_void_p), ("ldw2", C.c_int64),
        ("a_mode", C.c_int32),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Hout", C.c_int32),
        ("Wout", C.c_int32), ("stride", C.c_int32), ("sign", C.c_int32), ("upsample", C.c_int32), ("transposed", C.c_int32),
        ("shift", C.c_int32),
        ("alpha", C.c_float),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("rows_per_group", C.c_int64), ("ldrb", C.c_int64),
        ("R", C.c_void_p), ("ldr", C.c_int64), ("r_dtype", C.c_int32),
        ("act", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("c_dtype", C.c_int32),
        ("C2", C.c_void_p), ("ldc2", C.c_int64),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_eps", C.c_float),
        ("sync", C.c_void_p), ("sync_count", C.c_int64),
        ("split_out", C.POINTER(C.c_int32)),
        ("rs_out", C.c_void_p), ("rs_in", C.c_void_p), ("rs_ld", C.c_int64), ("rs_n", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("Sq", C.c_int32), ("Skv", C.c_int32), ("hd", C.c_int32), ("causal", C.c_int32),
        ("scale", C.c_float),
        ("Q", C.c_void_p), ("ldq", C.c_int64),
        ("K", C.c_void_p), ("ldk", C.c_int64),
        ("V", C.c_void_p), ("ldv", C.c_int64),
        ("O", C.c_void_p), ("ldo", C.c_int64),
        ("LSE", C.c_void_p),
        ("dO", C.c_void_p), ("lddo", C.c_int64),
        ("Delta", C.c_void_p),
        ("dQ", C.c_void_p), ("lddq", C.c_int64),
        ("dK", C.c_void_p), ("lddk", C.c_int64),
        ("dV", C.c_void_p), ("lddv", C.c_int64),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("fp8_ws", C.c_void_p), ("fp8_ws_bytes", C.c_int64),
    ]


class FfDesc(C.Structure):
    """tb_ff_desc (include/textboost_hip.h): the fused GEGLU feed-forward of the C = 320 transformer blocks"""
    _fields_ = [
        ("M", C.c_int64), ("C", C.c_int32), ("inner", C.c_int32),
        ("X", C.c_void_p), ("ldx", C.c_int64),
        ("W1", C.c_void_p), ("ldw1", C.c_int64),
        ("W2", C.c_void_p), ("ldw2", C.c_int64),
        ("b1", C.c_void_p), ("b2", C.c_void_p),
        ("HG", C.c_void_p), ("ldhg", C.c_int64),
        ("R", C.c_void_p), ("ldr", C.c_int64),
        ("Y", C.c_void_p), ("ldy", C.c_int64),
        ("ln_x", C.c_void_p), ("ld_lnx", C.c_int64), ("ln_stats", C.c_void_p), ("ln_gamma", C.c_void_p),
        # round 6, tb_ff_fwd only: attn2.to_out + residual + norm3 in front of the feed-forward, proj_out + residual behind it, same launch
        ("pre_W", C.c_void_p), ("ld_prew", C.c_int64), ("pre_b", C.c_void_p), ("pre_R", C.c_void_p), ("ld_prer", C.c_int64),
        ("pre_Y", C.c_void_p), ("ld_prey", C.c_int64), ("pre_gamma", C.c_void_p), ("pre_beta", C.c_void_p), ("pre_stats", C.c_void_p),
        ("pre_eps", C.c_float),
        ("post_W", C.c_void_p), ("ld_postw", C.c_int64), ("post_b", C.c_void_p), ("post_R", C.c_void_p), ("ld_postr", C.c_int64),
        ("post_Y", C.c_void_p), ("ld_posty", C.c_int64),
    ]


class ChainDesc(C.Structure):
    """tb_chain_desc (include/textboost_hip.h): Linear -> LayerNorm -> Linear on the C = 320 residual stream in one launch"""
    _fields_ = [
        ("M", C.c_int64),
        ("X", C.c_void_p), ("ldx", C.c_int64),
        ("W1", C.c_void_p), ("ldw1", C.c_int64), ("b1", C.c_void_p), ("R1", C.c_void_p), ("ldr1", C.c_int64),
        ("T", C.c_void_p), ("ldt", C.c_int64),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("stats", C.c_void_p),
        ("W2", C.c_void_p), ("ldw2", C.c_int64), ("N2", C.c_int32), ("b2", C.c_void_p),
        ("Y", C.c_void_p), ("ldy", C.c_int64),
    ]


class OptDesc(C.Structure):
    """tb_opt_desc (include/textboost_hip.h): the optimizer tail as two launches"""
    _fields_ = [
        ("state", C.c_void_p), ("grad", C.c_void_p),
        ("p_lora", C.c_void_p), ("m_lora", C.c_void_p), ("v_lora", C.c_void_p), ("n_lora", C.c_int64),
        ("p_added", C.c_void_p), ("m_emb", C.c_void_p), ("v_emb", C.c_void_p), ("n_added", C.c_int32), ("D", C.c_int32),
        ("p_unet", C.c_void_p), ("m_unet", C.c_void_p), ("v_unet", C.c_void_p), ("n_unet", C.c_int64),
        ("p_decay", C.c_void_p), ("n_decay", C.c_int64), ("decay_factor", C.c_float),
        ("added_norms", C.c_void_p),
        ("lr_table", C.c_void_p), ("lr_table_n", C.c_int32),
        ("lr", C.c_float), ("emb_lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("wd", C.c_float),
        ("max_norm", C.c_float), ("mean_norm", C.c_float),
        ("growth_factor", C.c_float), ("backoff_factor", C.c_float), ("growth_interval", C.c_float), ("use_scaler", C.c_int32),
        ("grad_div", C.c_float),
        ("ws", C.c_void_p),
    ]


_lib = None

_VP, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "tb_gemm": ([C.POINTER(GemmDesc), _VP], C.c_int),
    "tb_gemm_ln_epilogue_ok": ([C.c_int64, C.c_int64, C.c_int64], C.c_int),
    "tb_gemm_subpixel_ok": ([_I, _I, _I, _I, _I], C.c_int),
    "tb_gemm_lnfold_ok": ([_I64, _I64, C.POINTER(C.c_int)], C.c_int),
    "tb_ff_fused_ok": ([C.c_int64, _I, _I], C.c_int),
    "tb_ff_fwd": ([C.POINTER(FfDesc), _VP], C.c_int),
    "tb_ff_bwd": ([C.POINTER(FfDesc), _VP], C.c_int),
    "tb_ff_debug": ([_VP], C.c_int),
    "tb_chain320_ok": ([C.c_int64, _I], C.c_int),
    "tb_chain320": ([C.POINTER(ChainDesc), _VP], C.c_int),
    "tb_last_hip_error": ([], C.c_char_p),
    "tb_mfma_peak_probe": ([_VP, _I, _I, _VP], C.c_int),
    "tb_gemm_set_variant": ([_I], C.c_int),
    "tb_gemm_last_config": ([_VP], None),
    "tb_attention_set_variant": ([_I], C.c_int),
    "tb_groupnorm_set_variant": ([_I], C.c_int),
    "tb_attention_fp8_ws_bytes": ([_I, _I, _I], C.c_int64),
    "tb_gemm8_set": ([_I], C.c_int),
    "tb_gemm8_last": ([_VP], C.c_int),
    "tb_gemm8_debug": ([_VP], C.c_int),
    "tb_groupnorm_ws_floats": ([_I, _I, _I, _I], _I64),
    "tb_groupnorm_fwd": ([_VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _VP], C.c_int),
    "tb_groupnorm_bwd": ([_VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _I64, _VP, _I, _I, _I, _I, _I, _VP], C.c_int),
    "tb_groupnorm_splitk_ok": ([_I, _I, _I, _I], C.c_int),
    "tb_groupnorm_fwd_splitk": ([_VP, _I, _I64, _VP, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _VP], C.c_int),
    "tb_groupnorm_bwd_splitk": ([_VP, _I, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _I64, _I, _I, _I, _I, _I, _VP], C.c_int),
    "tb_layernorm_fwd": ([_VP, _I64, _I, _VP, _I64, _I, _VP, _VP, _VP, _I64, _I, _F, _VP], C.c_int),
    "tb_layernorm_lora_fwd": ([_VP, _I64, _I, _VP, _I64, _I, _VP, _VP, _VP, _I64, _I, _F, _VP, _I, _VP, _I64, _VP], C.c_int),
    "tb_layernorm_lora_rows_fwd": ([_VP, _I64, _I, _VP, _I64, _I, _VP, _VP, _VP, _I64, _I, _F, _VP, _I, _VP, _I64, _I64, _VP], C.c_int),
    "tb_layernorm_bwd": ([_VP, _I64, _I, _VP, _I64, _I, _VP, _VP, _VP, _I64, _VP, _I64, _VP, _I64, _I64, _I, _VP], C.c_int),
    "tb_attention_fwd": ([C.POINTER(AttnDesc), _VP], C.c_int),
    "tb_attention_bwd": ([C.POINTER(AttnDesc), _VP], C.c_int),
    "tb_add_noise": ([_VP, _VP, _VP, _VP, _VP, _VP, _I, _I64, _VP], C.c_int),
    "tb_timestep_embed": ([_VP, _VP, _I, _I, _VP], C.c_int),
    "tb_conv4_to_nhwc": ([_VP, _I, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_convin_to_nhwc": ([_VP, _I, _I, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_softmax_rows": ([_VP, _I64, _VP, _I64, _I64, _I, _VP], C.c_int),
    "tb_vae_sample": ([_VP, _I64, _VP, _VP, _I, _I, _I, _F, _VP], C.c_int),
    "tb_chan_mix": ([_VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP], C.c_int),
    "tb_dpm_step": ([_VP, _VP, _VP, _VP, _I64, _I, _F, _F, _F, _F, _F, _F, _VP], C.c_int),
    "tb_vae_image": ([_VP, _I64, _VP, _I, _I, _I, _VP], C.c_int),
    "tb_conv_to4": ([_VP, _I64, _VP, _VP, _VP, _I, _I, _I, _I, _VP], C.c_int),
    "tb_mse_loss": ([_VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP], C.c_int),
    "tb_kpl_cos": ([_VP, _I64, _VP, _I64, _I, _VP, _I64, _VP, _VP, _VP, _F, _I64, _I, _VP], C.c_int),
    "tb_kpl_mse": ([_VP, _I64, _VP, _I64, _I, _VP, _I64, _VP, _VP, _VP, _F, _I64, _I, _VP], C.c_int),
    "tb_geglu_bwd": ([_VP, _I64, _VP, _I64, _VP, _I64, _I64, _I, _VP], C.c_int),
    "tb_pool2x2_sum": ([_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP], C.c_int),
    "tb_upsample2x": ([_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP], C.c_int),
    "tb_add_f16": ([_VP, _I64, _VP, _I64, _VP, _I64, _I64, _I, _VP], C.c_int),
    "tb_convert": ([_VP, _I64, _I, _VP, _I64, _I, _I64, _I, _F, _VP], C.c_int),
    "tb_embed_fwd": ([_VP, _VP, _VP, _I, _VP, _I, _I64, _I, _I, _VP], C.c_int),
    "tb_embed_bwd": ([_VP, _VP, _VP, _I64, _I, _I64, _I, _VP], C.c_int),
    "tb_textboost_pin_fwd": ([_VP, _I, _VP, _VP, _I, _I, _I, _I, _I64, _VP], C.c_int),
    "tb_textboost_pin_bwd": ([_VP, _VP, _I, _I, _I, _I, _I64, _VP], C.c_int),
    "tb_lora_down": ([_VP, _I64, _VP, _VP, _I64, _I64, _I, _I, _VP], C.c_int),
    "tb_lora_pack": ([_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_lora_bwd_ws_floats": ([_I64, _I, _I, _I, _I], _I64),
    "tb_lora_bwd_chain": ([_VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _I64, _VP, _VP, _I64, _I, _I, _I, _I, _F, _VP, _I64, _VP, _I64, _VP, _I,
                           _VP], C.c_int),
    "tb_lora_set_variant": ([_I], C.c_int),
    "tb_boundary_conv_set_variant": ([_I], C.c_int),
    "tb_lora_bwd": ([_VP, _I64, _VP, _I64, _VP, _I64, _VP, _VP, _I64, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_sumsq": ([_VP, _I64, _VP, _VP, _VP], C.c_int),
    "tb_scaler_update": ([_VP, _F, _F, _F, _F, _F, _F, _I, _F, _VP], C.c_int),
    "tb_lr_from_table": ([_VP, _VP, _I, _VP], C.c_int),
    "tb_adamw": ([_VP, _VP, _VP, _VP, _I64, _F, _F, _F, _F, _F, _VP, _I, _VP], C.c_int),
    "tb_weight_decay": ([_VP, _I64, _F, _VP, _VP], C.c_int),
    "tb_renorm_rows": ([_VP, _I, _I, _F, _VP, _VP], C.c_int),
    "tb_row_norms": ([_VP, _I64, _I, _VP, _VP], C.c_int),
    "tb_optimizer_tail": ([C.POINTER(OptDesc), _VP], C.c_int),
    "tb_gemm_f32": ([C.POINTER(GemmDesc), _VP], C.c_int),
    "tb_gemm_f32_t": ([C.POINTER(GemmDesc), _I, _I, _VP], C.c_int),
    "tb_attention_f32_ws_floats": ([_I, _I, _I, _I], _I64),
    "tb_attention_f32_fwd": ([C.POINTER(AttnDesc), _VP, _I64, _VP], C.c_int),
    "tb_attention_f32_bwd": ([C.POINTER(AttnDesc), _VP, _I64, _VP], C.c_int),
    "tb_groupnorm_f32_fwd": ([_VP, _I64, _VP, _I64, _VP, _VP, _VP, _I, _I, _I, _I, _F, _I, _VP], C.c_int),
    "tb_groupnorm_f32_bwd": ([_VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _I64, _VP, _I64, _I, _I, _I, _I, _I, _VP], C.c_int),
    "tb_add_noise_f32": ([_VP, _VP, _VP, _VP, _VP, _VP, _I, _I64, _VP], C.c_int),
    "tb_timestep_embed_f32": ([_VP, _VP, _I, _I, _VP], C.c_int),
    "tb_conv4_to_nhwc_f32": ([_VP, _I, _VP, _VP, _VP, _I64, _I, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_conv_to4_f32": ([_VP, _I64, _VP, _VP, _VP, _I, _I, _I, _I, _VP], C.c_int),
    "tb_upsample2x_f32": ([_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP], C.c_int),
    "tb_pool2x2_sum_f32": ([_VP, _I64, _VP, _I64, _I, _I, _I, _I, _VP], C.c_int),
    "tb_add_f32": ([_VP, _I64, _VP, _I64, _VP, _I64, _I64, _I, _VP], C.c_int),
    "tb_lora_pack_f32": ([_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _F, _VP], C.c_int),
    "tb_kv_lora_pack_f32": ([_VP, _VP, _VP, _I64, _I, _I, _F, _VP], C.c_int),
    "tb_mse_loss_f32": ([_VP, _VP, _VP, _VP, _VP, _I64, _VP, _VP], C.c_int),
    "tb_resample_ksize": ([_I, _I, _I], C.c_int),
    "tb_resample_coeffs": ([_I, _I, _I, _VP, _VP], C.c_int),
    "tb_affine_nearest_tables": ([C.POINTER(C.c_double), _I, _I, _I, _I, _VP, _VP], C.c_int),
    "tb_img_resample": ([_VP, _I64, _I, _I, _VP, _I64, _I, _VP, _VP, _I, _I, _I, _VP], C.c_int),
    "tb_img_gather": ([_VP, _I64, _VP, _I64, _I, _I, _VP, _VP, _I, _VP], C.c_int),
    "tb_img_affine_bicubic": ([_VP, _I64, _I, _I, _I, _I, _VP, _I64, _I, _I, _I, _I, C.POINTER(C.c_double), _VP], C.c_int),
    "tb_img_to_pixels": ([_VP, _I64, _I, _I, _I, _I, _VP, _I, _VP], C.c_int),
    "tb_img_pack_rgb": ([_VP, _VP, _I64, _VP], C.c_int),
    "tb_img_unpack_rgb": ([_VP, _I64, _I, _I, _VP, _VP], C.c_int),
}
IMG_LANCZOS, IMG_BICUBIC = 1, 3


_half = "fp16"     # the 16-bit float of the ACTIVE library: one process runs one numeric mode (like the reference's accelerator)
_libs = {}


def set_half(kind: str):
    """select the library every `lib()` call returns: "fp16" (libtextboost_hip.so) or "bf16" (libtextboost_hip_bf16.so, same C-ABI, TB_F16 then
    means bfloat16).  Returns the previous kind."""
    global _half, _lib
    assert kind in ("fp16", "bf16"), kind
    prev, _half = _half, kind
    _lib = _libs.get(kind)
    return prev


@contextlib.contextmanager
def use_half(kind: str):
    """run a block on the other build of the library (host-side dispatch only: every op looks `lib()` up when it is called, so this also
    works while a HIP graph is being captured).  The VAE encoder uses it to stay on the fp16 build in a bf16 run (vae.py)."""
    if kind is None:   # "whatever the process runs"
        yield
        return
    prev = set_half(kind)
    try:
        yield
    finally:
        set_half(prev)


def half_kind() -> str:
    return _half


def half_dtype():
    """torch dtype of the active library's 16-bit float"""
    return torch.bfloat16 if _half == "bf16" else torch.float16


def lib():
    global _lib
    if _lib is None:
        path = LIB_PATH_BF16 if _half == "bf16" else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m textboost_amd.build` (hipcc --offload-arch=gfx950). "
                "textboost_amd has no CPU fallback.")
        _lib = C.CDLL(path)
        for name, (args, res) in _SIGS.items():
            fn = getattr(_lib, name)  # AttributeError here = library older than the header
            fn.argtypes, fn.restype = args, res
        _libs[_half] = _lib
    return _lib


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        detail = ""
        if rc == -5:
            detail = " (HIP: " + (lib().tb_last_hip_error() or b"?").decode() + ")"
        raise RuntimeError(f"{what} failed with code {rc}{detail}")
