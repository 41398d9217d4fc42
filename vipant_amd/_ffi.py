"""ctypes binding of libvipant_hip.so (the C ABI declared in include/vipant_hip.h).

The product path has NO CPU or eager-PyTorch fallback: if the library is missing or a call fails,
an exception is raised.  Build the library with `python -m vipant_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch must be imported before the library is loaded: libvipant_hip.so depends on libamdhip64.so and has to bind to
# the HIP runtime torch ships and initialises; loading the system copy first gives the process two runtimes and
# torch then reports "No HIP GPUs are available".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VIPANT_HIP_LIB") or os.path.join(_HERE, "lib", "libvipant_hip.so")   # override: A/B timing of builds

EPI_BF16, EPI_F32, EPI_RESIDUAL_F32, EPI_QUICKGELU, EPI_DQUICKGELU, EPI_SCALE_F32, EPI_QUICKGELU_D8, EPI_DQUICKGELU_D8 = range(8)
EPI_FEW_ROWS = 0x100          # VIPANT_EPI_FEW_ROWS: one row per item of a batch (read-out rows): the 64 x 64 split-K kernel
STREAM_FEW_ROWS = 0x100
STREAM_ACT_Q = 0x200
STREAM_IN_F16, STREAM_OUT_F16 = 1, 2          # VIPANT_STREAM_*: precision of the residual stream inside the transformer stack
LN_DY_F32, LN_DRES_BF16, LN_X_F16 = 1, 2, 4

_p, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t

class Fp8Plan(C.Structure):
    """struct vipant_fp8_plan (include/vipant_hip.h): e4m3 weights + activation scratch of one fused block operator call."""
    _fields_ = [("w_q", _p), ("w_scale", _p), ("w2_q", _p), ("w2_scale", _p), ("act_q", _p), ("act_scale", _p),
                ("emit_q", _p), ("emit_scale", _p), ("dy_q", _p), ("dy_scale", _p), ("tn_e4m3", _i64), ("keep_q", _p), ("keep_scale", _p),
                ("keep2_q", _p), ("keep2_scale", _p)]


# name -> (restype, argtypes); mirrors include/vipant_hip.h one to one
PROTOTYPES = {
    "vipant_last_error": (C.c_char_p, []),
    "vipant_version": (_i32, []),
    "vipant_device_check": (_i32, []),
    "vipant_gemm_nt": (_i32, [_p, _i64, _p, _i64, _p, _i64, _p, _p, _f32, _i64, _i64, _i64, _i32, _p]),
    "vipant_gemm_tn_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vipant_gemm_tn": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "vipant_gemm_nt_tokens": (_i32, [_p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "vipant_tokens_cls_rows": (_i32, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_gemm_tn_pair_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vipant_gemm_tn_pair": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "vipant_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "vipant_colsum_bf16": (_i32, [_p, _i64, _p, _i64, _i64, _i32, _p, _sz, _p]),
    "vipant_layernorm_fwd": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p]),
    "vipant_residual_add": (_i32, [_p, _p, _p, _i64, _i32, _p]),
    "vipant_layernorm_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "vipant_layernorm_fwd_e4m3": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _i32, _p]),
    "vipant_layernorm_bwd_e4m3": (_i32, [_p, _i32, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i32, _i64, _i64, _p, _sz, _p, _p, _p]),
    "vipant_layernorm_bwd": (_i32, [_p, _i32, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i32, _i64, _i64, _p, _sz, _p]),
    "vipant_mha_fwd": (_i32, [_p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mha_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mha_fwd_e4m3": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mha_bwd_e4m3": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mha_rows_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mha_rows_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_rows_ctx_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _p]),
    "vipant_rows_ctx_bwd": (_i32, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _p]),
    "vipant_gemm_nt_heads": (_i32, [_p, _i64, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "vipant_head_expand": (_i32, [_p, _p, _i64, _i64, _p]),
    "vipant_head_extract": (_i32, [_p, _i32, _p, _p, _i64, _i64, _p]),
    "vipant_gather_rows_bytes": (_i32, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_add_rows_bf16": (_i32, [_p, _p, _p, _i32, _i64, _i64, _i64, _p]),
    "vipant_cast_bf16_multi": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _i64, _p]),
    "vipant_cast_f32": (_i32, [_p, _p, _i64, _p]),
    "vipant_quant_e4m3_rows": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "vipant_gemm_nt_e4m3": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_mx_scale_bytes": (_sz, [_i64, _i64]),
    "vipant_quant_e4m3_mx": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "vipant_quant_e4m3_mx_cols": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "vipant_quant_e4m3_mx32": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _p]),
    "vipant_mx_uniform32": (_i32, [_p, _i64, _p, _i64, _i64, _p]),
    "vipant_quant_e4m3_mx32_cols": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "vipant_mx_uniform32_cols": (_i32, [_p, _i64, _p, _i64, _i64, _i64, _i64, _p]),
    "vipant_gemm_tn_e4m3_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vipant_gemm_tn_e4m3": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "vipant_cast_bf16": (_i32, [_p, _p, _p, _i64, _i64, _p]),
    "vipant_conv_weight_prep": (_i32, [_p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_im2col": (_i32, [_p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p]),
    "vipant_assemble_tokens": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_assemble_tokens_bwd": (_i32, [_p, _p, _p, _p, _i32, _i64, _i64, _i64, _p]),
    "vipant_conv_weight_grad": (_i32, [_p, _p, _i64, _i64, _i64, _i32, _p]),
    "vipant_l2norm_fwd": (_i32, [_p, _p, _p, _i64, _i64, _p]),
    "vipant_l2norm_bwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _p]),
    "vipant_embed_tokens": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_gather_rows": (_i32, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_scatter_rows": (_i32, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_scatter_rows_bf16": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_infonce_workspace_bytes": (_sz, [_i64, _i64]),
    "vipant_infonce_strip_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vipant_infonce_fwd_bwd": (_i32, [_p, _p, _p, _f32, _p, _p, _p, _p, _f32, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "vipant_retrieval_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vipant_retrieval_ranks": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _sz, _p]),
    "vipant_fbank_workspace_bytes": (_sz, [_i64]),
    "vipant_fbank": (_i32, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _f32, _p, _sz, _p]),
    # fused operator set (vipant_amd/csrc/block.hip)
    "vipant_block_workspace_bytes": (_sz, [_i64, _i64]),
    "vipant_ln_qkv_fwd": (_i32, [_p] * 11 + [_i64, _i64, _p]),
    "vipant_ln_qkv_fwd_e4m3": (_i32, [_p] * 11 + [_i64, _i64, _p, _i32, _p]),
    "vipant_ln_qkv_bwd_e4m3": (_i32, [_p] * 15 + [_i64, _i64, _p, _sz, _p, _i32, _p]),
    "vipant_gemm_bias_residual_fwd_e4m3": (_i32, [_p] * 5 + [_i64, _i64, _i64, _p, _p]),
    "vipant_gemm_bias_residual_bwd_e4m3": (_i32, [_p] * 5 + [_i64, _i64, _i64, _p, _sz, _p, _p]),
    "vipant_ln_mlp_quickgelu_fwd_e4m3": (_i32, [_p] * 15 + [_i64, _i64, _p, _i32, _p]),
    "vipant_mlp_quickgelu_recompute_e4m3": (_i32, [_p] * 5 + [_i64, _i64, _p, _p]),
    "vipant_ln_mlp_quickgelu_bwd_e4m3": (_i32, [_p] * 20 + [_i64, _i64, _p, _sz, _p, _i32, _p]),
    "vipant_ln_qkv_bwd": (_i32, [_p] * 15 + [_i64, _i64, _p, _sz, _p]),
    "vipant_gemm_bias_residual_fwd": (_i32, [_p] * 5 + [_i64, _i64, _i64, _p]),
    "vipant_gemm_bias_residual_bwd": (_i32, [_p] * 5 + [_i64, _i64, _i64, _p, _sz, _p]),
    "vipant_ln_mlp_quickgelu_fwd": (_i32, [_p] * 15 + [_i64, _i64, _p]),
    "vipant_mlp_quickgelu_recompute": (_i32, [_p] * 5 + [_i64, _i64, _p]),
    "vipant_ln_mlp_quickgelu_bwd": (_i32, [_p] * 20 + [_i64, _i64, _p, _sz, _p]),
    "vipant_patch_embed_ln_fwd": (_i32, [_p] * 13 + [_i64] * 10 + [_i32, _p]),
    "vipant_patch_embed_ln_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "vipant_patch_embed_ln_bwd": (_i32, [_p, _i32] + [_p] * 13 + [_i64] * 5 + [_i32, _p, _sz, _p]),
    "vipant_cls_ln_proj_l2norm_fwd": (_i32, [_p] * 12 + [_i64] * 4 + [_i32, _p]),
    "vipant_cls_ln_proj_l2norm_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "vipant_cls_ln_proj_l2norm_bwd": (_i32, [_p] * 18 + [_i64] * 4 + [_i32, _p, _sz, _p]),
    "vipant_embed_gather_pos_fwd": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "vipant_eot_ln_proj_l2norm_fwd": (_i32, [_p] * 12 + [_i64] * 4 + [_i32, _p]),
    "vipant_lars_workspace_bytes": (_sz, [_i64]),
    "vipant_lars_step": (_i32, [_p, _p, _p, _p, _p, _p, _i64, _f32, _f32, _f32, _p, _sz, _p]),
}


class VipantError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VipantError(
                f"{LIB_PATH} not found: the HIP hot path is not built (run `python -m vipant_amd.build`); "
                "there is no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        older = "VIPANT_HIP_LIB" in os.environ       # an A/B run against another (possibly older) build: its missing entry points fail at use
        for name, (res, args) in PROTOTYPES.items():
            if older and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)      # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def call(name: str, *args):
    """Invoke a status-returning entry point; raise VipantError with the library's message on failure."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().vipant_last_error().decode(errors="replace")
        raise VipantError(f"{name} failed with code {rc}: {msg}")


def query(name: str, *args) -> int:
    return int(getattr(lib(), name)(*args))
