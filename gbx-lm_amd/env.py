"""The ONE reader of the package's environment switches (host side; the kernels' switches are csrc/gbx_env.h).

Every switch is a row of ``TABLE`` -- name, default, what it does -- and ``get`` refuses names that are not in it, so a switch
cannot exist without its line of documentation (DESIGN.md section 8 is this table plus the C++ one).  None is needed for normal use.
"""
from __future__ import annotations

import os

TABLE = {
    # name: (default, doc)
    "GBX_QLINEAR_LIB": ("", "path of the product shared library (default gbx-lm_amd/csrc/libgbx_qlinear.so): experiment / stamps builds"),
    "GBX_DGEMV": ("1", "FusedDecodeRunner: the four GEMV launches of a block on gbx_dgemv_forward (0: gbx_qlinear_fused_forward) (A/B)"),
    "GBX_DG_SSQ": ("1", "FusedDecodeRunner: carry the RMSNorm statistic from the o_proj / down_proj launch to the next norm'd launch (A/B)"),
    "GBX_QKV_CAT": ("1", "FusedDecodeRunner: q|k|v as one matrix concatenated at load time (A/B)"),
    "GBX_QKV_SLICES": ("1", "DecodeRunner (B > 1): q|k|v as ONE k-split GEMM whose fp32 slices the attention launch sums (0: three segments, one launch, no split) (A/B)"),
    "GBX_DATTN": ("1", "FusedDecodeRunner: norm + q|k|v + attention + o_proj as ONE launch with XCD-local hand-offs (gbx_dattn_forward; 8 kv heads of 128, unsplit context): three launches per block instead of five (0: the five-launch block) (A/B)"),
    "GBX_NORM_XCD": ("1", "batched decode: the slice-sum + residual + RMSNorm launches spread over the whole chip, the row statistic exchanged inside one XCD's L2 (gbx_add_rmsnorm_slices_xcd; 0: one workgroup per row) (A/B)"),
    "GBX_DMLP": ("1", "FusedDecodeRunner, with GBX_DATTN: gate|up + SwiGLU + down_proj as ONE launch with an XCD-local hand-off (gbx_dmlp_forward; 2 bits, group 128, bf16, Llama-3-8B MLP shapes): two launches per block instead of three (0: gate|up and down as two launches) (A/B)"),
    "GBX_DMLP_MX3": ("1", "FusedDecodeRunner, with GBX_DMLP: 3-bit blocks (group 128, Llama-3 shapes) on the fused MLP launch in its matrix-core form, their rows repacked once as a 2-bit and a 1-bit plane (gbx_repack_planes; +3/8 byte per weight of device memory); 0: the gate|up and down launches; 2: also for isolated 3-bit blocks of a mixed model, where it measured -1.5 % (A/B; the library reads the same switch)"),
    "GBX_DMLP_MX4": ("1", "FusedDecodeRunner, with GBX_DMLP: 4-bit blocks (group 128, Llama-3-8B shapes) on the fused MLP launch in its matrix-core form (a 4-bit field = two 2-bit fields of one nibble: dmlp.hip); 0: the gate|up and down launches (A/B; the library reads the same switch to choose between the matrix-core and the vector form)"),
    "GBX_DMLP_VEC": ("0", "FusedDecodeRunner, with GBX_DMLP: the fused MLP launch for 3- and 4-bit blocks too (dmlpv.hip: vector-ALU dots, down_proj's words through LDS; 8B and 3.2-3B shapes).  Off by default: measured 1.5-2.5 % slower than the gate|up and down launches it replaces (profiles/r06_dmlpv.txt) (A/B)"),
    "GBX_DBLOCK": ("1", "FusedDecodeRunner, with GBX_DATTN and GBX_DMLP: the whole 2-bit block as ONE launch with a chip-wide release inside and the MLP's rows streamed into LDS during the attention chain (gbx_dblock_forward; Llama-3-8B shapes, contexts up to gbx_dblock_max_context(); 0: two launches per block) (A/B)"),
    "GBX_FUSE_AO": ("0", "FusedDecodeRunner: attention + o_proj as one launch (libgbx_experimental.so; measured neutral)"),
    "GBX_GEMV_MXD": ("0", "route the fused GEMV launches through libgbx_experimental.so's one-row-per-operand scaled-MFMA GEMV (1: all eligible, 2: gate|up only)"),
    "GBX_GEMV_MX": ("0", "route the fused GEMV launches through libgbx_experimental.so's loader/consumer scaled-MFMA GEMV"),
    "GBX_PROMPT_GRAPH": ("1", "graph-replayed prompt buckets for 2..128-token prompts (0: eager pass)"),
    "GBX_DENSE_CACHE_GB": ("32", "budget of the resident dense weight images of long prompts, GiB (0 disables)"),
    "GBX_TP_GRAPH": ("1", "bench.py: RCCL collectives inside the HIP graph (0: eager launches)"),
    "GBX_TP_P2P": ("0", "bench.py: one-shot P2P all-reduce instead of RCCL (falls back to RCCL, with the reason reported, where it cannot work)"),
}


def get(name: str) -> str:
    """Value of switch ``name`` (its table default when unset); KeyError for an undocumented name."""
    default, _ = TABLE[name]
    return os.environ.get(name, default)


def flag(name: str) -> bool:
    return get(name) not in ("", "0")
