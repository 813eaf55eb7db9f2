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


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(gbx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ("gbx_version", "gbx_last_error", "gbx_qlinear_forward", "gbx_qlinear_workspace_bytes",
                 "gbx_qlinear_check", "gbx_unpack", "gbx_dequantize"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"


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
