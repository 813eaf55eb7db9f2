"""ctypes binding of the C-ABI in include/gbx_qlinear.h (libgbx_qlinear.so, gfx950 HIP kernels).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The CPU
oracle under ``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
from . import env as _env  # noqa: E402
_SO = _env.get("GBX_QLINEAR_LIB") or os.path.join(_CSRC, "libgbx_qlinear.so")   # override: experiment builds only

GBX_OK, GBX_ERR_INVALID, GBX_ERR_HIP, GBX_ERR_UNSUPPORTED = 0, 1, 2, 3
GBX_F16, GBX_BF16, GBX_F32 = 0, 1, 2
GBX_STREAM_OP_BYTES = 160

_XSO = os.path.join(_CSRC, "libgbx_experimental.so")        # opt-in kernels behind include/gbx_experimental.h
_lib = None
_xlib = None


def build(force: bool = False, jobs: int = 8) -> str:
    """Compile every HIP source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return _SO


def so_path() -> str:
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                f"gbx_lm_amd: HIP extension {_SO} is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C gbx-lm_amd/csrc`). There is no CPU fallback for QuantizedLinear.")
        # PyTorch's HIP runtime must be initialised BEFORE this library is mapped: loaded first, the library's fat binaries register
        # with a runtime that torch then never initialises for them, and every launch fails with "no ROCm-capable device is detected"
        # (seen with `build(); smoke()` in one process; any process that imports torch.cuda first was fine).  No GPU: a cheap False.
        import torch
        torch.cuda.is_available()
        L = ctypes.CDLL(_SO)
        i64, i32, p = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
        L.gbx_version.restype = ctypes.c_int
        L.gbx_last_error.restype = ctypes.c_char_p
        L.gbx_qlinear_check.argtypes = [i64, i32, i32, i32, i32]
        L.gbx_qlinear_workspace_bytes.argtypes = [i64, i32, i32, i32, i32]
        L.gbx_qlinear_workspace_bytes.restype = ctypes.c_size_t
        L.gbx_qlinear_forward.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, p]
        L.gbx_qlinear_forward_ws.argtypes = [p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, p, ctypes.c_size_t, p]
        L.gbx_unpack.argtypes = [p, p, i32, i32, i32, p]
        L.gbx_dequantize.argtypes = [p, p, p, p, i32, i32, i32, i32, i32, i32, p]
        L.gbx_logprobs_workspace_bytes.restype = ctypes.c_size_t
        if hasattr(L, 'gbx_qlinear_partials_bytes'):
            L.gbx_qlinear_partials_bytes.argtypes = [i64, i32, i32, i32, i32]
            L.gbx_qlinear_partials_bytes.restype = ctypes.c_size_t
        L.gbx_attn_workspace_bytes.restype = ctypes.c_size_t
        if hasattr(L, 'gbx_dblock_state_bytes'):
            L.gbx_dblock_state_bytes.argtypes = []
            L.gbx_dblock_state_bytes.restype = ctypes.c_size_t
        if hasattr(L, 'gbx_norm_state_bytes'):
            L.gbx_norm_state_bytes.argtypes = []
            L.gbx_norm_state_bytes.restype = ctypes.c_size_t
        if hasattr(L, 'gbx_dattn_state_bytes'):
            L.gbx_dattn_state_bytes.argtypes = []
            L.gbx_dattn_state_bytes.restype = ctypes.c_size_t
        if hasattr(L, 'gbx_p2p_inbox_bytes'):
            L.gbx_p2p_inbox_bytes.argtypes = [i32, i32]
            L.gbx_p2p_inbox_bytes.restype = ctypes.c_size_t
        L.gbx_dense_swiglu_forward.argtypes = [p, p, p, i64, i32, i32, i32, p]
        L.gbx_dense_workspace_bytes.argtypes = [i64, i32, i32]
        L.gbx_dense_workspace_bytes.restype = ctypes.c_size_t
        L.gbx_dense_forward_ws.argtypes = [p, p, p, p, i64, i32, i32, i32, p, ctypes.c_size_t, p]
        L.gbx_qlinear_swiglu_workspace_bytes.argtypes = [i64, i32, i32]
        L.gbx_qlinear_swiglu_workspace_bytes.restype = ctypes.c_size_t
        L.gbx_qlinear_swiglu_forward_ws.argtypes = [p, p, p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, p, ctypes.c_size_t, p]
        _declare_optional(L)
        if L.gbx_version() < GBX_MIN_VERSION and not _env.get("GBX_QLINEAR_LIB"):
            raise RuntimeError(f"gbx_lm_amd: {_SO} reports ABI version {L.gbx_version()}, this binding needs >= {GBX_MIN_VERSION}: "
                               "rebuild it (`make -C gbx-lm_amd/csrc`)")
        _lib = L
    return _lib


class FusedDesc(ctypes.Structure):
    """Mirror of ``gbx_qlinear_fused`` (include/gbx_qlinear.h)."""
    _fields_ = [("x", ctypes.c_void_p),
                ("K", ctypes.c_int32), ("bits", ctypes.c_int32), ("group_size", ctypes.c_int32), ("act_dtype", ctypes.c_int32),
                ("nseg", ctypes.c_int32),
                ("qweight", ctypes.c_void_p * 3), ("scales", ctypes.c_void_p * 3), ("zeros", ctypes.c_void_p * 3),
                ("y", ctypes.c_void_p * 3), ("N", ctypes.c_int32 * 3),
                ("norm_weight", ctypes.c_void_p), ("norm_eps", ctypes.c_float),
                ("residual", ctypes.c_void_p), ("swiglu", ctypes.c_int32),
                ("sz", ctypes.c_void_p)]


class ExpFusedDesc(ctypes.Structure):
    """Mirror of ``gbx_exp_qlinear_fused`` (include/gbx_experimental.h): FusedDesc + the fields of the launch behind
    gbx_attn_oproj_decode."""
    _fields_ = FusedDesc._fields_ + [("x_acc", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("zero_buf", ctypes.c_void_p),
                                     ("zero_n", ctypes.c_int32)]


def exp_lib() -> ctypes.CDLL:
    """libgbx_experimental.so (opt-in / measured-negative kernels; never needed for normal use)."""
    global _xlib
    if _xlib is None:
        if not os.path.exists(_XSO):
            raise RuntimeError(f"gbx_lm_amd: {_XSO} is not built (`make -C gbx-lm_amd/csrc`)")
        lib()                                       # (initialisation order: see lib())
        L = ctypes.CDLL(_XSO)
        i32, p, f32 = ctypes.c_int32, ctypes.c_void_p, ctypes.c_float
        L.gbx_exp_version.restype = ctypes.c_int
        L.gbx_exp_last_error.restype = ctypes.c_char_p
        L.gbx_exp_qlinear_fused_forward.argtypes = [ctypes.POINTER(ExpFusedDesc), p]
        L.gbx_stream_encode.argtypes = [ctypes.POINTER(ExpFusedDesc), i32, p, ctypes.c_size_t]
        L.gbx_stream_run.argtypes = [p, p, i32, i32, i32, p]
        L.gbx_attn_oproj_supported.argtypes = [i32, i32, i32, i32, i32, i32]
        L.gbx_attn_oproj_decode.argtypes = [p, p, p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, f32, f32, i32, i32, p]
        L.gbx_qlinear_prepack_tiles.argtypes = [p, p, p, p, p, p, p, p, i32, i32, i32, i32, p]
        L.gbx_gemv_tiled.argtypes = [p, p, p, p, i32, i32, i32, i32, p, f32, p, i32, i32, p]
        if hasattr(L, 'gbx_exp_batch_mx_forward'):
            L.gbx_exp_batch_mx_image_bytes.argtypes = [i32]
            L.gbx_exp_batch_mx_image_bytes.restype = ctypes.c_size_t
            L.gbx_exp_batch_mx_aux_bytes.argtypes = [i32]
            L.gbx_exp_batch_mx_aux_bytes.restype = ctypes.c_size_t
            L.gbx_exp_batch_mx_image.argtypes = [p, ctypes.c_int64, i32, p, p, p]
            L.gbx_exp_batch_mx_forward.argtypes = [p, p, p, p, p, p, p, ctypes.c_int64, i32, i32, p]
            L.gbx_exp_batch_mx_slices.argtypes = [i32, i32]
        _xlib = L
    return _xlib


def check_exp(rc: int) -> None:
    if rc == GBX_OK:
        return
    msg = exp_lib().gbx_exp_last_error().decode("utf-8", "replace")
    if rc == GBX_ERR_INVALID:
        raise ValueError(msg)
    if rc == GBX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


class DgDesc(ctypes.Structure):
    """Mirror of ``gbx_dgemv`` (include/gbx_qlinear.h)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("epilogue", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("qweight", ctypes.c_void_p), ("qweight_up", ctypes.c_void_p), ("sz", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("norm_weight", ctypes.c_void_p),
                ("ssq_in", ctypes.c_void_p), ("ssq_out", ctypes.c_void_p),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("bits", ctypes.c_int32), ("group_size", ctypes.c_int32),
                ("act_dtype", ctypes.c_int32), ("rows_per_batch", ctypes.c_int32), ("norm_eps", ctypes.c_float),
                ("x_acc", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("zero_buf", ctypes.c_void_p), ("zero_n", ctypes.c_int32)]


class DattnDesc(ctypes.Structure):
    """Mirror of ``gbx_dattn`` (include/gbx_qlinear.h): the fused attention launch of a decoder block at M == 1."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("act_dtype", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("norm_weight", ctypes.c_void_p), ("ssq_in", ctypes.c_void_p),
                ("qkv_qweight", ctypes.c_void_p), ("qkv_sz", ctypes.c_void_p), ("kcache", ctypes.c_void_p), ("vcache", ctypes.c_void_p),
                ("inv_freq", ctypes.c_void_p), ("pos", ctypes.c_void_p), ("o_qweight", ctypes.c_void_p), ("o_scales", ctypes.c_void_p),
                ("o_zeros", ctypes.c_void_p), ("acc", ctypes.c_void_p), ("state", ctypes.c_void_p), ("attn_out", ctypes.c_void_p),
                ("x_acc", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("zero_buf", ctypes.c_void_p), ("zero_n", ctypes.c_int32),
                ("D", ctypes.c_int32), ("n_heads", ctypes.c_int32), ("capacity", ctypes.c_int32),
                ("qkv_bits", ctypes.c_int32), ("qkv_group_size", ctypes.c_int32), ("o_bits", ctypes.c_int32), ("o_group_size", ctypes.c_int32),
                ("traditional", ctypes.c_int32), ("norm_eps", ctypes.c_float), ("rope_scale", ctypes.c_float), ("sm_scale", ctypes.c_float)]


class DmlpDesc(ctypes.Structure):
    """Mirror of ``gbx_dmlp`` (include/gbx_qlinear.h): the fused MLP launch of a decoder block at M == 1."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("act_dtype", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("x_acc", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("norm_weight", ctypes.c_void_p),
                ("gate_qweight", ctypes.c_void_p), ("up_qweight", ctypes.c_void_p), ("gate_up_sz", ctypes.c_void_p),
                ("down_qweight", ctypes.c_void_p), ("down_sz", ctypes.c_void_p), ("acc_out", ctypes.c_void_p),
                ("zero_buf", ctypes.c_void_p), ("state", ctypes.c_void_p), ("act_out", ctypes.c_void_p),
                ("D", ctypes.c_int32), ("I", ctypes.c_int32), ("zero_n", ctypes.c_int32), ("norm_eps", ctypes.c_float),
                ("gate_up_bits", ctypes.c_int32), ("gate_up_group_size", ctypes.c_int32), ("down_bits", ctypes.c_int32),
                ("down_group_size", ctypes.c_int32), ("weight_layout", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class DblockDesc(ctypes.Structure):
    """Mirror of ``gbx_dblock`` (include/gbx_qlinear.h): a whole 2-bit decoder block at M == 1 as one launch."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("act_dtype", ctypes.c_int32),
                ("x", ctypes.c_void_p), ("norm_weight", ctypes.c_void_p), ("ssq_in", ctypes.c_void_p),
                ("qkv_qweight", ctypes.c_void_p), ("qkv_sz", ctypes.c_void_p), ("kcache", ctypes.c_void_p), ("vcache", ctypes.c_void_p),
                ("inv_freq", ctypes.c_void_p), ("pos", ctypes.c_void_p), ("o_qweight", ctypes.c_void_p), ("o_scales", ctypes.c_void_p),
                ("o_zeros", ctypes.c_void_p), ("acc_o", ctypes.c_void_p), ("x_acc", ctypes.c_void_p), ("x_out", ctypes.c_void_p),
                ("attn_out", ctypes.c_void_p),
                ("norm2_weight", ctypes.c_void_p), ("gate_qweight", ctypes.c_void_p), ("up_qweight", ctypes.c_void_p),
                ("gate_up_sz", ctypes.c_void_p), ("down_qweight", ctypes.c_void_p), ("down_sz", ctypes.c_void_p),
                ("acc_d", ctypes.c_void_p), ("x2_out", ctypes.c_void_p), ("act_out", ctypes.c_void_p),
                ("zero_buf0", ctypes.c_void_p), ("zero_buf1", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("zero_n0", ctypes.c_int32), ("zero_n1", ctypes.c_int32), ("capacity", ctypes.c_int32), ("traditional", ctypes.c_int32),
                ("norm_eps", ctypes.c_float), ("norm2_eps", ctypes.c_float), ("rope_scale", ctypes.c_float), ("sm_scale", ctypes.c_float)]


GBX_DG_PLAIN, GBX_DG_RESIDUAL, GBX_DG_SWIGLU = 0, 1, 2
GBX_DG_SSQ_SLOTS = 1024
GBX_LAYOUT_REFERENCE, GBX_LAYOUT_PLANES = 0, 1
GBX_MIN_VERSION = 1302          # the header this binding mirrors


def _declare_optional(L) -> None:
    """argtypes of the entry points beyond the core forward (declared one by one so that a diagnostic build that lacks
    one of them -- GBX_QLINEAR_LIB -- still loads; calling a missing entry point raises AttributeError)."""
    i64, i32, p, f32 = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_float
    table = {
        "gbx_qlinear_fused_forward": [ctypes.POINTER(FusedDesc), p],
        "gbx_dgemv_plan": [i32, i32, i32, i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)],
        "gbx_dgemv_pack": [p, p, p, p, p, i32, i32, i32, p],
        "gbx_dgemv_forward": [ctypes.POINTER(DgDesc), p],
        "gbx_dgemv_kernel_id": [ctypes.POINTER(DgDesc), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)],
        "gbx_dattn_plan": [i32, i32, i32, i32, i32, i32, i32, i32],
        "gbx_dattn_forward": [ctypes.POINTER(DattnDesc), p],
        "gbx_dattn_status": [p, ctypes.POINTER(i32)],
        "gbx_dmlp_plan": [i32, i32, i32, i32, i32, i32, i32],
        "gbx_dmlp_forward": [ctypes.POINTER(DmlpDesc), p],
        "gbx_acc_residual": [p, p, p, i32, i32, p, p, p, p],
        "gbx_xcd_layout_check": [],
        "gbx_dblock_plan": [i32] * 14,
        "gbx_dblock_forward": [ctypes.POINTER(DblockDesc), p],
        "gbx_dblock_status": [p, ctypes.POINTER(i32)],
        "gbx_dblock_max_context": [],
        "gbx_qlinear_kernel_id": [i64, i32, i32, i32, i32, i32],
        "gbx_qlinear_swiglu_forward": [p, p, p, p, p, p, p, p, i64, i32, i32, i32, i32, i32, p],
        "gbx_qlinear_forward_partials": [p, p, p, p, p, ctypes.c_size_t, ctypes.POINTER(i32), i64, i32, i32, i32, i32, i32, p],
        "gbx_add_rmsnorm_slices": [p, p, i32, p, p, i64, i32, f32, i32, p],
        "gbx_add_rmsnorm_slices_xcd": [p, p, i32, p, p, i64, i32, f32, i32, p, p],
        "gbx_dense_forward": [p, p, p, p, i64, i32, i32, i32, p],
        "gbx_attn_decode": [p, p, p, p, p, p, p, p, i32, i32, i32, i32, f32, f32, i32, i32, p],
        "gbx_attn_decode_batched": [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, f32, i32, i32, p],
        "gbx_attn_decode_normed": [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, f32, i32, p, p, f32, p, p, p, i32, p],
        "gbx_attn_decode_ws": [p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, f32, i32, p, p, f32, p, p, p, i32, p, ctypes.c_size_t, p],
        "gbx_attn_decode_slices": [p, i32, p, p, p, p, p, i32, i32, i32, i32, i32, f32, f32, i32, p, p, f32, i32, p, ctypes.c_size_t, p],
        "gbx_attn_workspace_bytes": [i32, i32, i32, i32, i32],
        "gbx_attn_cached": [p, p, p, p, p, i32, i32, i32, i32, i32, f32, i32, p],
        "gbx_rope_kv_write": [p, p, p, p, p, i64, i64, i32, i32, i32, i64, i64, i64, i64, i64, i64, i32, i32, f32, i32, i32, p],
        "gbx_attn_prefill": [p, p, p, p, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i64, i32, f32, i32, p],
        "gbx_moe_route": [p, p, p, i64, i32, i32, i32, i32, i32, p],
        "gbx_moe_combine": [p, p, p, i64, i32, i32, i32, p],
        "gbx_moe_combine_residual": [p, p, p, p, i64, i32, i32, i32, p],
        "gbx_kv_append": [p, p, p, p, p, i64, i32, i32, i32, i32, p],
        "gbx_logprobs_argmax": [p, p, p, p, i32, i32, p],
        "gbx_logprobs_argmax_packed": [p, p, p, p, i32, i32, p, p, p, p, p],
        "gbx_repack_planes": [p, p, i64, i32, i32, p],
        "gbx_dmlp_weight_layout": [i32, i32, i32, i32, i32, i32, i32],
        "gbx_add_rmsnorm": [p, p, p, p, i64, i32, f32, i32, p],
        "gbx_swiglu": [p, p, p, i64, i32, p],
        "gbx_dense_gemv": [p, p, p, p, f32, i32, i32, i32, p],
        "gbx_rmsnorm": [p, p, p, i64, i32, f32, i32, p],
        "gbx_dense_gemv_argmax": [p, p, p, p, f32, i32, i32, p, i32, p],
        "gbx_p2p_inbox_create": [i32, i32, ctypes.POINTER(p), p],
        "gbx_p2p_inbox_open": [p, ctypes.POINTER(p)],
        "gbx_p2p_inbox_close": [p],
        "gbx_p2p_status": [p, i32, i32, ctypes.POINTER(i32)],
        "gbx_p2p_inbox_destroy": [p],
        "gbx_p2p_allreduce": [p, i32, i32, i32, ctypes.POINTER(p), i32, i32, p],
        "gbx_dense_gemv_argmax_shard": [p, p, p, p, f32, i32, i32, p, i32, i32, p],
        "gbx_embed_next_token_sharded": [p, i32, p, p, p, p, p, i32, i32, i32, i32, p],
        "gbx_embed_next_token": [p, p, p, p, p, i32, i32, i32, i32, i32, p],
        "gbx_qlinear_forward_multi": [p, i32, p, p, p, p, p, p, i64, i32, i32, i32, i32, p],
        "gbx_gather_qlinear_forward": [p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
        "gbx_gather_swiglu_forward": [p, p, p, p, p, p, p, p, p, i32, i32, i32, i32, i32, i32, i32, i32, p],
        "gbx_eminf_optimize": [p, i32, p, p, p, i32, f32, f32, i32, p],
        "gbx_rope": [p, p, p, p, i64, i64, i32, i32, i32, i32, i64, i64, i64, f32, f32, i32, i32, p],
        "gbx_rope_append": [p, p, p, p, p, p, p, i32, i32, i32, i32, i32, f32, i32, i32, p],
        "gbx_pack_scales_zeros": [p, p, p, p, p, i32, i32, p],
    }
    for name, argtypes in table.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes = argtypes


def check(rc: int) -> None:
    """Map a C return code to the reference's exception behaviour."""
    if rc == GBX_OK:
        return
    msg = lib().gbx_last_error().decode("utf-8", "replace")
    if rc == GBX_ERR_INVALID:
        raise ValueError(msg)
    if rc == GBX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
