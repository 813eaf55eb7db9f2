"""CPU oracle for the GBA QuantizedLinear hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``gbx_lm_amd``) never imports anything from ``oracle/`` and fails
loudly when the HIP library is missing.

PARITY STATUS: **parity unpinned** for the bit order inside a packed row.
The reference delegates the arithmetic to ``mx.quantized_matmul`` (third-party
``mlx>=0.25.0`` -- requirements.txt:2, setup.py:12 -- not vendored, not installable
here) and none of its tests feed known bytes through the op.
What IS pinned by reference code and restated line by line below:

  * tensor shapes / dtypes        gbx_lm/models/quantized_linear_gba.py:60-117
  * forward wiring, "zeros added" gbx_lm/models/quantized_linear_gba.py:151-155,195-205
  * double-quant expansion        gbx_lm/models/quantized_linear_gba.py:119-149
  * loader dtype casts            gbx_lm/utils.py:841-843
  * strategy -> (bits, group)     gbx_lm/models/quantized_linear_gba.py:231-276

The one external assumption (MLX affine packing = little-endian, LSB-first
bit-stream per row) is isolated in :func:`unpack`.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

try:  # torch is only needed for bf16 helpers and the double-quant restatement
    import torch
except Exception:  # pragma: no cover
    torch = None

VALID_BITS = (2, 3, 4)
VALID_GROUPS = (32, 64, 128)


# --------------------------------------------------------------------------- #
# integer part: bit-exact                                                      #
# --------------------------------------------------------------------------- #
def unpack(qweight: np.ndarray, bits: int) -> np.ndarray:
    """qweight [N, K*bits/32] uint32  ->  q [N, K] int32.

    Field k of row n is the ``bits``-wide field at bit offset k*bits of the row's
    little-endian bit-stream (the MLX affine format that
    ``mx.quantized_matmul`` at quantized_linear_gba.py:195-203 consumes).
    Implemented by literal bit-stream expansion so that straddling 3-bit fields
    need no special case.
    """
    if bits not in VALID_BITS:
        raise ValueError(f"bits must be one of {VALID_BITS}, got {bits}")
    qw = np.ascontiguousarray(qweight, dtype="<u4")
    n, w = qw.shape
    if (w * 32) % bits:
        raise ValueError("row length is not a whole number of fields")
    k = w * 32 // bits
    bitstream = np.unpackbits(qw.view(np.uint8).reshape(n, w * 4), axis=1, bitorder="little")
    fields = bitstream.reshape(n, k, bits).astype(np.int32)
    weights = (1 << np.arange(bits, dtype=np.int32))
    return (fields * weights).sum(axis=2, dtype=np.int32)


def unpack_formula(qweight: np.ndarray, bits: int) -> np.ndarray:
    """Same as :func:`unpack` via the word/shift formula of SURVEY.md Appendix A
    (independent second derivation; tests assert both agree)."""
    qw = np.ascontiguousarray(qweight, dtype=np.uint32).astype(np.uint64)
    n, w = qw.shape
    k = w * 32 // bits
    kk = np.arange(k, dtype=np.int64)
    o = kk * bits
    wi = o >> 5
    s = (o & 31).astype(np.uint64)
    mask = np.uint64((1 << bits) - 1)
    lo = (qw[:, wi] >> s) & mask
    spill = (s + np.uint64(bits)) > 32
    wi2 = np.minimum(wi + 1, w - 1)
    hi = (qw[:, wi2] << (np.uint64(32) - s)) & mask
    return np.where(spill[None, :], lo | hi, lo).astype(np.int32)


def pack(q: np.ndarray, bits: int) -> np.ndarray:
    """Inverse of :func:`unpack`: q [N, K] ints in [0, 2^bits) -> uint32 [N, K*bits/32]."""
    q = np.asarray(q)
    n, k = q.shape
    if (k * bits) % 32:
        raise ValueError("K*bits must be a multiple of 32")
    if q.min() < 0 or q.max() >= (1 << bits):
        raise ValueError("field out of range")
    fields = ((q[:, :, None].astype(np.int64) >> np.arange(bits)) & 1).astype(np.uint8)
    by = np.packbits(fields.reshape(n, k * bits), axis=1, bitorder="little")
    return np.ascontiguousarray(by).view("<u4").reshape(n, k * bits // 32).astype(np.uint32)


# --------------------------------------------------------------------------- #
# floating-point part                                                          #
# --------------------------------------------------------------------------- #
def _to_f32(a) -> np.ndarray:
    """Accept numpy (f16/f32/f64), torch (incl. bf16) -> float32 numpy."""
    if torch is not None and isinstance(a, torch.Tensor):
        return a.detach().to(torch.float32).cpu().numpy()
    return np.asarray(a, dtype=np.float32)


def dequant(qweight, scales, zeros, bits: int, group_size: int, dtype=np.float32) -> np.ndarray:
    """w[n,k] = scales[n,k//g] * q[n,k] + zeros[n,k//g]   ("plus": quantized_linear_gba.py:151-155)."""
    if group_size not in VALID_GROUPS:
        raise ValueError(f"The group size value ({group_size}) must be 32, 64 or 128.")
    q = unpack(np.asarray(qweight), bits)
    n, k = q.shape
    if k % group_size:
        raise ValueError("K must be a multiple of group_size")
    s = _to_f32(scales).astype(dtype)
    z = _to_f32(zeros).astype(dtype)
    if s.shape != (n, k // group_size) or z.shape != s.shape:
        raise ValueError(f"scales/zeros must be [{n},{k // group_size}], got {s.shape}/{z.shape}")
    s = np.repeat(s, group_size, axis=1)
    z = np.repeat(z, group_size, axis=1)
    return (q.astype(dtype) * s + z).astype(dtype)


def qlinear_ref(x, qweight, scales, zeros, bits: int, group_size: int, bias=None,
                acc=np.float32) -> np.ndarray:
    """y[..., n] = sum_k x[..., k] * w[n, k] (+ bias[n])   quantized_linear_gba.py:195-205.

    ``acc=np.float64`` gives the y_ref64 used as the tolerance anchor.
    Returns ``acc``-typed numpy; callers round to the activation dtype themselves.
    """
    w = dequant(qweight, scales, zeros, bits, group_size, dtype=np.float32).astype(acc)
    xf = _to_f32(x).astype(acc)
    y = xf @ w.T
    if bias is not None:
        y = y + _to_f32(bias).astype(acc)
    return y


def gather_qlinear_ref(x, qweight, scales, zeros, indices, bits: int, group_size: int, bias=None,
                       acc=np.float32) -> np.ndarray:
    """Expert-indexed QuantizedLinear (QuantizedSwitchLinear.__call__, gbx_lm/models/switch_layers.py:78-92):
    ``mx.gather_qmm(x, qweight[E,N,W], scales[E,N,G], zeros, rhs_indices=indices, transpose=True)`` + ``bias[indices]``.

    x: [..., 1, 1, K] (one row shared by the k expert slots, as SwitchGLU feeds up/gate, switch_layers.py:176) or
    [..., k, 1, K] (one row per slot, as it feeds down_proj); indices: [..., k] -> y [..., k, 1, N] with
    y[..., j, 0, :] = W[indices[..., j]] . x[..., j or 0, 0, :].  ``sorted_indices`` changes nothing numerically."""
    idx = np.asarray(indices)
    xf = _to_f32(x).astype(acc)
    qw = np.asarray(qweight)
    E, N = qw.shape[0], qw.shape[1]
    K = xf.shape[-1]
    k = idx.shape[-1]
    lead = idx.shape[:-1]
    xb = np.broadcast_to(xf, lead + (k, 1, K)).reshape(-1, K)
    flat = idx.reshape(-1)
    y = np.zeros((flat.shape[0], N), dtype=acc)
    for e in np.unique(flat):
        if not 0 <= int(e) < E:
            raise IndexError(f"expert index {int(e)} out of range [0, {E})")
        w = dequant(qw[e], _to_f32(scales)[e], _to_f32(zeros)[e], bits, group_size, dtype=np.float32).astype(acc)
        rows = flat == e
        y[rows] = xb[rows] @ w.T
        if bias is not None:
            y[rows] += _to_f32(bias)[e].astype(acc)
    return y.reshape(lead + (k, 1, N))


def round_to(y: np.ndarray, act_dtype: str):
    """Round an oracle result to the activation dtype ('bf16' | 'f16'); returns torch tensor."""
    t = torch.from_numpy(np.asarray(y, dtype=np.float32))
    return t.to(torch.bfloat16 if act_dtype == "bf16" else torch.float16)


def tolerance(y_ref64: np.ndarray, act_dtype: str):
    """Stated fp tolerance: |y - y_ref64| <= atol + rtol*|y_ref64|."""
    rms = float(np.sqrt(np.mean(np.square(np.asarray(y_ref64, dtype=np.float64))))) if y_ref64.size else 0.0
    if act_dtype == "bf16":
        return 2e-2, 2e-2 * rms
    return 4e-3, 4e-3 * rms


def assert_close(y, y_ref64, act_dtype: str, what: str = ""):
    rtol, atol = tolerance(y_ref64, act_dtype)
    y = _to_f32(y).astype(np.float64)
    err = np.abs(y - y_ref64)
    bound = atol + rtol * np.abs(y_ref64)
    bad = err > bound
    if bad.any():
        i = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.size} outside tolerance (rtol={rtol}, atol={atol:.3e}); "
            f"worst at {i}: got {y[i]:.6g} want {y_ref64[i]:.6g}")


# --------------------------------------------------------------------------- #
# GBA double-quant expansion (fully specified by reference code)               #
# --------------------------------------------------------------------------- #
def create_scales_zeros(qstatistic, qzeros_zeros, qzeros_scales, qscales_zeros, qscales_scales,
                        input_dims: int, output_dims: int, group_size: int):
    """Restates quantized_linear_gba.py:119-149 (torch-CPU fp16 arithmetic, then transpose to [N, G]).

    qstatistic u8 [ceil(K/g), ceil(N/32), 32]; the four fp16 side tensors [ceil(K/g), ceil(N/32), 1].
    Returns (scales, zeros) fp16 torch tensors shaped [N, K/g].
    """
    qs = torch.as_tensor(np.asarray(qstatistic)).to(torch.uint8)
    f = lambda a: torch.as_tensor(np.asarray(a)).to(torch.float16)
    buffer_shape = (math.ceil(input_dims / group_size), output_dims)            # :133
    qscales = (qs & 0xF0) >> 4                                                  # :134
    qzeros = qs & 0x0F                                                          # :135
    zeros = ((qzeros.to(torch.float16) - f(qzeros_zeros)) * f(qzeros_scales)).view(buffer_shape)   # :137
    scales = ((qscales.to(torch.float16) - f(qscales_zeros)) * f(qscales_scales)).view(buffer_shape)  # :138
    return scales.t().contiguous(), zeros.t().contiguous()                     # :141-142


# --------------------------------------------------------------------------- #
# strategy lookup (per-layer mixed bpw)                                        #
# --------------------------------------------------------------------------- #
_STRATEGY_KEYS = ['kv_a_proj_with_mqa', 'kv_b_proj', 'q_a_proj', 'q_b_proj', 'q_proj', 'k_proj',
                  'v_proj', 'o_proj', 'gate_proj', 'up_proj', 'down_proj', 'qkv_proj', 'gate_up_proj']


def strategy_lookup(name: str, strategy: dict, switch: bool = False):
    """(bits, group_size) for module ``name``; restates quantized_linear_gba.py:256-272
    including the substring-order quirk and the KeyError fall-through."""
    layer_number = name.split('.')[2]
    per_block = strategy["model.layers.{}".format(layer_number)]
    strg = None
    if switch:                 # QuantizedSwitchLinear: 'moe_expert_' + key over gate/up/down only (:238-246)
        for key in ['gate_proj', 'up_proj', 'down_proj']:
            if key in name:
                try:
                    strg = per_block['moe_expert_' + key]
                    break
                except KeyError:
                    pass
    else:
        for key in _STRATEGY_KEYS:
            if key in name:
                try:
                    if "shared_expert" in name:
                        key = 'moe_shared_expert_' + key
                    strg = per_block[key]
                    break
                except KeyError:
                    pass
    if strg is None:
        raise KeyError(f"no strategy entry matches {name}")
    bits = strg["bits"][0]
    group_size = strg["group_size"][str(bits)]
    assert group_size in [32, 64, 128], f"The group size value ({group_size}) must be 32, 64 or 128."
    return bits, group_size


# --------------------------------------------------------------------------- #
# EMINF logits optimiser (second caller of the path; pure function)            #
# --------------------------------------------------------------------------- #
def eminf_optimize(logits, alpha=0.65, num_steps=None, threshold=0.05):
    """numpy fp32 restatement of gbx_lm/infer_opt.py:30-77 (same update rule, same stopping tests)."""
    f32 = np.float32

    def softmax(x):
        e = np.exp(x - x.max(), dtype=f32)
        return (e / e.sum(dtype=f32)).astype(f32)

    def entropy(x):
        p = softmax(x)
        return f32(-np.sum(p * np.log(p + f32(1e-10), dtype=f32), dtype=f32))

    cur = np.asarray(logits, dtype=f32)
    a = alpha
    h0 = entropy(cur)
    best_h, best = h0, cur
    target = max(threshold, 0.1 * float(h0))
    if num_steps is None:
        r = max(0.0, float(h0) - threshold) / threshold
        num_steps = int(3 + 15 * (r / (1.0 + r)))
    for _ in range(num_steps):
        h = entropy(cur)
        if float(h) < target:
            break
        z = cur - cur.max()
        logp = (z - np.log(np.sum(np.exp(z, dtype=f32), dtype=f32), dtype=f32)).astype(f32)
        p = np.exp(logp, dtype=f32)
        grad = p * (np.sum(p * (logp + f32(1.0)), dtype=f32) - (logp + f32(1.0)))
        gn = np.sqrt(np.sum(grad * grad, dtype=f32), dtype=f32)
        if float(gn) > 0:
            new = (cur - f32(a) * (grad / gn)).astype(f32)
            hn = entropy(new)
            if float(hn) < float(h):
                cur = new
                if float(hn) < float(best_h):
                    best, best_h = new, hn
            else:
                a *= 0.65
                if a < 1e-4:
                    break
        else:
            break
    return best, best_h


# --------------------------------------------------------------------------- #
# synthetic inputs                                              #
# --------------------------------------------------------------------------- #
def synth_layer(n: int, k: int, bits: int, group_size: int, seed: int = 0, act_dtype: str = "bf16",
                m: int = 1, bias: bool = False):
    """Deterministic synthetic packed weights + activations with the SURVEY 8d distributions."""
    rng = np.random.default_rng(seed)
    qweight = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
    g = k // group_size
    qmax = (1 << bits) - 1
    sc = rng.uniform(0.5, 1.5, size=(n, g)) * (2.0 / qmax) * 0.02
    ze = -sc * qmax / 2.0 + rng.normal(0.0, 1.0, size=(n, g)) * 0.1 * sc
    tdt = torch.bfloat16 if act_dtype == "bf16" else torch.float16
    scales = torch.from_numpy(sc.astype(np.float32)).to(tdt)
    zeros = torch.from_numpy(ze.astype(np.float32)).to(tdt)
    x = torch.from_numpy(rng.normal(0.0, 1.0, size=(m, k)).astype(np.float32)).to(tdt)
    b = torch.from_numpy(rng.normal(0.0, 0.1, size=(n,)).astype(np.float32)).to(tdt) if bias else None
    return dict(qweight=qweight, scales=scales, zeros=zeros, x=x, bias=b)
