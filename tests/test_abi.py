"""The C-ABI library loads on a GPU-less box and exports every symbol include/*.h declares.
No compute calls here (those are the -m gpu tests)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gbx_lm_amd import _lib
    if not os.path.exists(_lib.so_path()):
        _lib.build()
    return _lib.lib()


def declared_symbols(header="gbx_qlinear.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gbx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ("gbx_version", "gbx_last_error", "gbx_qlinear_forward", "gbx_qlinear_workspace_bytes",
                 "gbx_qlinear_check", "gbx_unpack", "gbx_dequantize"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/gbx_qlinear.h but not exported"


def test_experimental_library_is_separate():
    """The opt-in kernels live behind their own header and library: every symbol of include/gbx_experimental.h is exported by
    libgbx_experimental.so and by nothing else; the product header does not declare them."""
    from gbx_lm_amd import _lib
    if not os.path.exists(_lib._XSO):
        _lib.build()
    xl = _lib.exp_lib()
    prod = set(declared_symbols())
    exp = declared_symbols("gbx_experimental.h")
    assert exp and "gbx_exp_qlinear_fused_forward" in exp and "gbx_stream_run" in exp
    for name in exp:
        assert hasattr(xl, name), f"{name} declared in include/gbx_experimental.h but not exported by libgbx_experimental.so"
        assert name not in prod, f"{name} is declared by both headers"
        assert not hasattr(_lib.lib(), name), f"{name} leaked into the product library"
    assert xl.gbx_exp_version() >= 1100
    assert len(prod) <= 79, "the product ABI is what a gbx-lm maintainer binds, not the lab notebook"


def test_version_and_host_side_validation(lib):
    from gbx_lm_amd import _lib
    assert lib.gbx_version() >= 1000
    assert lib.gbx_qlinear_check(1, 4096, 4096, 4, 128) == _lib.GBX_OK
    for (n, k, b, g) in [(8, 256, 5, 64), (8, 256, 4, 48), (8, 250, 4, 32), (0, 256, 4, 32), (8, 96, 4, 64)]:
        assert lib.gbx_qlinear_check(1, n, k, b, g) == _lib.GBX_ERR_INVALID
        assert len(lib.gbx_last_error()) > 0
    with pytest.raises(ValueError, match="must be 32, 64 or 128"):
        _lib.check(lib.gbx_qlinear_check(1, 8, 256, 4, 48))
    assert lib.gbx_qlinear_workspace_bytes(1, 4096, 4096, 4, 128) == 0


def test_glue_entry_points_validate_arguments_on_the_host(lib):
    """Empty inputs succeed, malformed ones return GBX_ERR_INVALID / UNSUPPORTED with a message -- all before any HIP call."""
    from gbx_lm_amd import _lib
    one = ctypes.c_void_p(16)                                   # a non-null, 16-byte aligned stand-in (never dereferenced)
    assert lib.gbx_swiglu(None, None, None, 0, _lib.GBX_BF16, None) == _lib.GBX_OK
    assert lib.gbx_swiglu(one, one, one, 12, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID          # n % 8
    assert lib.gbx_swiglu(one, one, one, 16, _lib.GBX_F32, None) == _lib.GBX_ERR_INVALID           # activation dtype
    assert lib.gbx_add_rmsnorm(None, None, None, None, 0, 4096, 1e-5, _lib.GBX_BF16, None) == _lib.GBX_OK
    assert lib.gbx_add_rmsnorm(one, None, one, one, 4, 4096, 1e-5, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID
    assert lib.gbx_rmsnorm(None, None, None, 0, 128, 1e-5, _lib.GBX_BF16, None) == _lib.GBX_OK
    assert lib.gbx_rope(None, None, None, None, 0, 0, 8, 1, 128, 128, 0, 0, 0, 1.0, 1.0, 0, _lib.GBX_BF16, None) == _lib.GBX_OK
    assert lib.gbx_rope(one, one, one, None, 0, 1, 8, 1, 128, 128, 1024, 128, 128, 1.0, 1.0, 0, _lib.GBX_BF16, None) \
        == _lib.GBX_ERR_INVALID                                                                    # in place
    assert b"out of place" in lib.gbx_last_error()
    two = ctypes.c_void_p(32)
    assert lib.gbx_rope(one, two, one, None, 0, 1, 8, 1, 128, 130, 1024, 128, 128, 1.0, 1.0, 0, _lib.GBX_BF16, None) \
        == _lib.GBX_ERR_INVALID                                                                    # dims > D
    args = (one, one, one, one, one, one, one, one)
    assert lib.gbx_attn_decode_batched(*args, 0, 8, 2, 128, 64, 1.0, 0.1, 0, _lib.GBX_BF16, None) == _lib.GBX_OK
    assert lib.gbx_attn_decode_batched(*args, 2, 8, 2, 96, 64, 1.0, 0.1, 0, _lib.GBX_BF16, None) == _lib.GBX_ERR_UNSUPPORTED
    assert lib.gbx_attn_decode_batched(*args, 2, 8, 3, 128, 64, 1.0, 0.1, 0, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID
    with pytest.raises(NotImplementedError, match="head_dim"):
        _lib.check(lib.gbx_attn_decode_batched(*args, 2, 8, 2, 96, 64, 1.0, 0.1, 0, _lib.GBX_BF16, None))
    # the slice form of the batched attention (round 4): q | k | v as fp32 slices of one k-split GEMM
    sl = (one, one, one, one, one)                                 # kcache, vcache, inv_freq, pos, out
    tail = (1.0, 0.1, 0, None, None, 0.0, _lib.GBX_BF16, None, 0, None)
    assert lib.gbx_attn_decode_slices(one, 4, *sl, 0, 8, 2, 128, 64, *tail) == _lib.GBX_OK             # empty batch
    assert lib.gbx_attn_decode_slices(one, 0, *sl, 2, 8, 2, 128, 64, *tail) == _lib.GBX_ERR_UNSUPPORTED   # slices
    assert lib.gbx_attn_decode_slices(one, 9, *sl, 2, 8, 2, 128, 64, *tail) == _lib.GBX_ERR_UNSUPPORTED
    assert lib.gbx_attn_decode_slices(one, 4, *sl, 2, 8, 2, 96, 64, *tail) == _lib.GBX_ERR_UNSUPPORTED    # head_dim
    assert lib.gbx_attn_decode_slices(one, 4, *sl, 2, 8, 3, 128, 64, *tail) == _lib.GBX_ERR_INVALID       # n_heads % n_kv
    assert lib.gbx_attn_decode_slices(None, 4, *sl, 2, 8, 2, 128, 64, *tail) == _lib.GBX_ERR_INVALID      # null slices
    assert lib.gbx_attn_decode_slices(one, 4, *sl, 2, 8, 2, 128, 64, 1.0, 0.1, 0, one, None, 0.0, _lib.GBX_BF16, None, 0, None) \
        == _lib.GBX_ERR_INVALID                                                                    # q norm without k norm


def test_round2_entry_points_validate_arguments_on_the_host(lib):
    """The prompt-pass entry points added in round 2: sizes / shapes are answered and malformed calls refused before any HIP call."""
    from gbx_lm_amd import _lib
    one = ctypes.c_void_p(256)
    # dense path dispatch: always from 4096 rows; from 1024 rows when the 256 x 256 tiles fill the chip or the k range can be split
    assert lib.gbx_qlinear_workspace_bytes(32768, 4096, 4096, 4, 128) == 4096 * 4096 * 2
    assert lib.gbx_qlinear_workspace_bytes(1024, 14336, 4096, 4, 128) == 14336 * 4096 * 2            # 224 tiles: one round
    assert lib.gbx_qlinear_workspace_bytes(2048, 4096, 4096, 4, 128) == 4096 * 4096 * 2 + 2 * 2048 * 4096 * 4   # + 2 fp32 k slices
    assert lib.gbx_qlinear_kernel_id(2048, 4096, 4096, 4, 128, 1) in (31, 32, 33) and lib.gbx_qlinear_kernel_id(2048, 4096, 4096, 4, 128, 0) in (11, 12, 13)
    assert lib.gbx_qlinear_kernel_id(1000, 4096, 4096, 4, 128, 1) == 12 and lib.gbx_qlinear_kernel_id(1000, 1024, 4096, 4, 128, 1) == 15
    assert lib.gbx_dense_workspace_bytes(2048, 4096, 4096) == 2 * 2048 * 4096 * 4 and lib.gbx_dense_workspace_bytes(16384, 4096, 4096) == 0
    assert lib.gbx_qlinear_swiglu_workspace_bytes(16384, 14336, 4096) == 2 * 14336 * 4096 * 2
    assert lib.gbx_qlinear_swiglu_workspace_bytes(1024, 14336, 4096) == 2 * 14336 * 4096 * 2
    assert lib.gbx_qlinear_swiglu_workspace_bytes(1024, 1024, 4096) == 0 and lib.gbx_qlinear_swiglu_workspace_bytes(16384, 1000, 4096) == 0
    # refusals
    assert lib.gbx_dense_forward_ws(one, one, None, one, 4096, 4096, 4100, _lib.GBX_BF16, None, 0, None) == _lib.GBX_ERR_UNSUPPORTED      # K % 64
    assert lib.gbx_dense_forward_ws(one, None, None, one, 4096, 4096, 4096, _lib.GBX_BF16, None, 0, None) == _lib.GBX_ERR_INVALID
    assert lib.gbx_dense_forward_ws(None, None, None, None, 0, 4096, 4096, _lib.GBX_BF16, None, 0, None) == _lib.GBX_OK
    assert lib.gbx_dense_swiglu_forward(one, one, one, 4096, 1000, 4096, _lib.GBX_BF16, None) == _lib.GBX_ERR_UNSUPPORTED                  # N % 128
    assert lib.gbx_dense_swiglu_forward(one, one, one, 4096, 1024, 4096, _lib.GBX_F32, None) == _lib.GBX_ERR_INVALID
    a = (one, one, one, one)
    assert lib.gbx_attn_prefill(*a, 0, 8, 2, 16, 16, 128, 0, 0, 128, 0, 0, 0, 0, 0, 0.1, _lib.GBX_BF16, None) == _lib.GBX_OK               # empty batch
    assert lib.gbx_attn_prefill(*a, 1, 8, 2, 16, 16, 96, 0, 0, 96, 0, 0, 0, 0, 0, 0.1, _lib.GBX_BF16, None) == _lib.GBX_ERR_UNSUPPORTED     # head_dim
    assert lib.gbx_attn_prefill(*a, 1, 8, 3, 16, 16, 128, 0, 0, 128, 0, 0, 0, 0, 0, 0.1, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID       # heads
    assert lib.gbx_attn_prefill(*a, 1, 8, 2, 16, 8, 128, 0, 0, 128, 0, 0, 0, 0, 0, 0.1, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID        # S < offset + L
    assert lib.gbx_attn_prefill(*a, 1, 8, 2, 16, 16, 128, 0, 0, 132, 0, 0, 0, 0, 0, 0.1, _lib.GBX_BF16, None) == _lib.GBX_ERR_INVALID       # stride % 8


def test_product_path_never_imports_oracle():
    """The shipped package must not reference oracle/ (a product path through the oracle voids parity)."""
    pkg = os.path.join(ROOT, "gbx-lm_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{fn} imports oracle"
                assert "gba_oracle" not in src, f"{fn} mentions the oracle"


def test_forward_fails_loudly_without_gpu():
    import torch
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    q = QuantizedLinear(256, 8, False, 64, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        q(torch.zeros(1, 256, dtype=torch.bfloat16))
