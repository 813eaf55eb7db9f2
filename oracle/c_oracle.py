"""ctypes binding of oracle/gba_oracle.c.  TEST INFRASTRUCTURE ONLY (see gba_oracle.py header)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgba_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gba_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, i32, p = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.gba_oracle_unpack.argtypes = [p, i64, i64, i32, p]
        L.gba_oracle_dequant.argtypes = [p, p, p, i64, i64, i32, i32, i32, p]
        L.gba_oracle_qlinear.argtypes = [p, p, p, p, p, p, i64, i64, i64, i32, i32, i32, i32, i32]
        _lib = L
    return _lib


def _u16(t):
    """torch bf16/f16 tensor or numpy f16 -> (uint16 numpy view, dtype code 0=f16 1=bf16)."""
    import torch
    if isinstance(t, torch.Tensor):
        code = 1 if t.dtype == torch.bfloat16 else 0
        if t.dtype not in (torch.bfloat16, torch.float16):
            raise TypeError(f"expected bf16/f16, got {t.dtype}")
        return np.ascontiguousarray(t.contiguous().view(torch.int16).numpy().view(np.uint16)), code
    a = np.ascontiguousarray(t)
    if a.dtype != np.float16:
        raise TypeError("numpy input must be float16")
    return a.view(np.uint16), 0


def unpack(qweight: np.ndarray, bits: int) -> np.ndarray:
    qw = np.ascontiguousarray(qweight, dtype=np.uint32)
    n, w = qw.shape
    k = w * 32 // bits
    out = np.empty((n, k), dtype=np.int32)
    rc = lib().gba_oracle_unpack(qw.ctypes.data, n, k, bits, out.ctypes.data)
    if rc:
        raise ValueError(f"gba_oracle_unpack rc={rc}")
    return out


def dequant(qweight, scales, zeros, bits: int, group: int) -> np.ndarray:
    qw = np.ascontiguousarray(qweight, dtype=np.uint32)
    n, w = qw.shape
    k = w * 32 // bits
    s, sdt = _u16(scales)
    z, zdt = _u16(zeros)
    assert sdt == zdt
    out = np.empty((n, k), dtype=np.float32)
    rc = lib().gba_oracle_dequant(qw.ctypes.data, s.ctypes.data, z.ctypes.data, n, k, bits, group, sdt,
                                  out.ctypes.data)
    if rc:
        raise ValueError(f"gba_oracle_dequant rc={rc}")
    return out


def qlinear(x, qweight, scales, zeros, bits: int, group: int, bias=None, acc64: bool = False) -> np.ndarray:
    qw = np.ascontiguousarray(qweight, dtype=np.uint32)
    n, w = qw.shape
    k = w * 32 // bits
    xv, adt = _u16(x)
    xv = xv.reshape(-1, k)
    m = xv.shape[0]
    s, sdt = _u16(scales)
    z, _ = _u16(zeros)
    bptr = None
    if bias is not None:
        bv, _ = _u16(bias)
        bptr = bv.ctypes.data
    y = np.empty((m, n), dtype=np.float64)
    rc = lib().gba_oracle_qlinear(xv.ctypes.data, qw.ctypes.data, s.ctypes.data, z.ctypes.data, bptr,
                                  y.ctypes.data, m, n, k, bits, group, adt, sdt, int(acc64))
    if rc:
        raise ValueError(f"gba_oracle_qlinear rc={rc}")
    return y
