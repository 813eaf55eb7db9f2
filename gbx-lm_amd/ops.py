"""Python-side operator wrappers: torch tensors in, raw device pointers across the C-ABI.

``quantized_matmul`` mirrors the one MLX call the reference makes on this path
(``mx.quantized_matmul(x, w, scales=, biases=, transpose=True, group_size=, bits=)``,
gbx_lm/models/quantized_linear_gba.py:195-203): same argument names and meaning.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, env

_DT = {torch.float16: _lib.GBX_F16, torch.bfloat16: _lib.GBX_BF16, torch.float32: _lib.GBX_F32}


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"gbx_lm_amd: `{name}` lives on {t.device}; the GBA QuantizedLinear path runs only as HIP kernels on an "
            "MI355X (no CPU fallback).")


class NotCapturable(RuntimeError):
    """Raised by an op that would need a host round trip while a HIP graph is being captured (the only failure the graph-replay
    decode path treats as "decode eagerly instead")."""


class _DenseCache:
    """Dequantised copies of packed weights that prompt-sized calls on the dense path (M >= 4096 rows, or >= 1024 with enough tiles)
    left in their workspace, kept resident and reused
    by the next long call on the same tensors -- the second chunk of a prompt, the next request -- instead of being rebuilt
    (25 us per 4096 x 4096 matrix, 11 ms of an 8 x 4096-token prompt pass).  288 GB of HBM per GPU make the 16 GB an 8B model's
    bf16 image takes a fair trade; ``GBX_DENSE_CACHE_GB`` bounds it (default 32, 0 disables), least recently used entries go
    first.  An entry is valid only for the very tensor OBJECTS it was built from (weak references + their in-place version
    counters and storage addresses), so reloaded, re-created or modified weights never see a stale image; it is dropped the
    moment one of those tensors dies (finalizers), and a consumer on another stream waits for the event recorded after the build.
    NOT seen: a write through ``tensor.data`` (it bumps no version counter) -- call ``dense_cache_invalidate(tensor)`` after one."""

    def __init__(self):
        import collections
        self.budget = int(float(env.get("GBX_DENSE_CACHE_GB")) * (1 << 30))
        self.entries = collections.OrderedDict()
        self.nbytes = 0

    @staticmethod
    def _key(tensors, extra):
        return tuple(id(t) for t in tensors) + tuple(extra)

    def get(self, tensors, extra):
        key = self._key(tensors, extra)
        ent = self.entries.get(key)
        if ent is None:
            return None
        refs, versions, dense, event, stream = ent
        if any(r() is not t for r, t in zip(refs, tensors)) or versions != tuple((t._version, t.data_ptr()) for t in tensors):
            self._drop_key(key)
            return None
        self.entries.move_to_end(key)
        cur = torch.cuda.current_stream(dense.device)
        if event is not None and cur != stream:       # built on another stream: order this consumer behind the build
            cur.wait_event(event)
        return dense

    def _drop_key(self, key):
        ent = self.entries.pop(key, None)
        if ent is not None:
            self.nbytes -= ent[2].numel() * ent[2].element_size()

    def drop(self, tensors, extra):
        self._drop_key(self._key(tensors, extra))

    def invalidate(self, tensor):
        """Drop every image built from ``tensor`` (by identity)."""
        tid = id(tensor)
        for key in [k for k, ent in self.entries.items() if any(r() is tensor for r in ent[0]) or tid in k[:len(ent[0])]]:
            self._drop_key(key)

    def put(self, tensors, extra, dense):
        import weakref
        size = dense.numel() * dense.element_size()
        if size > self.budget or torch.cuda.is_current_stream_capturing():
            return
        key = self._key(tensors, extra)
        self._drop_key(key)
        while self.entries and self.nbytes + size > self.budget:
            _, old = self.entries.popitem(last=False)
            self.nbytes -= old[2].numel() * old[2].element_size()
        stream = torch.cuda.current_stream(dense.device)
        event = torch.cuda.Event()
        event.record(stream)
        self.entries[key] = (tuple(weakref.ref(t) for t in tensors), tuple((t._version, t.data_ptr()) for t in tensors), dense,
                             event, stream)
        self.nbytes += size
        for t in tensors:                              # a dead weight must not pin its image until LRU pressure finds it
            weakref.finalize(t, self._drop_key, key)

    def clear(self):
        self.entries.clear()
        self.nbytes = 0


_dense_cache = _DenseCache()


def dense_cache_clear() -> None:
    """Free the resident dequantised weight images (see ``_DenseCache``)."""
    _dense_cache.clear()


def dense_cache_invalidate(tensor: torch.Tensor) -> None:
    """Drop the resident images built from ``tensor`` (a packed weight, its scales or zeros).  Needed only after a write that
    PyTorch's version counter does not see (``tensor.data.copy_(...)``, a raw pointer write); in-place ops on the tensor or on a
    view of it, and re-assigned tensors, are detected without it."""
    _dense_cache.invalidate(tensor)


def dense_cache_prewarm(model, dtype: Optional[torch.dtype] = None) -> int:
    """Build the resident bf16 / fp16 image of every QuantizedLinear weight of ``model`` now (serving: the first request then already
    runs its prompt on dense GEMMs -- from 256 rows on -- instead of paying the dequantisation or the fused tiles).  Costs 2 bytes per
    weight of HBM under the ``GBX_DENSE_CACHE_GB`` budget; returns the bytes resident afterwards."""
    from .models.quantized_linear_gba import QuantizedLinear
    L = _lib.lib()
    for mod in model.modules():
        if not isinstance(mod, QuantizedLinear) or mod.scales is None or not mod.qweight.is_cuda:
            continue
        dt = dtype or mod.scales.dtype
        n, k = mod.output_dims, mod.input_dims
        if dt not in _DT or mod.scales.dtype != dt or k % 128 or n * k * 2 > _dense_cache.budget - _dense_cache.nbytes:
            continue
        src = (mod.qweight, mod.scales, mod.zeros)
        ckey = (n, k, mod.bits, mod.group_size, dt, mod.qweight.device.index)
        if not all(t.is_contiguous() for t in src) or _dense_cache.get(src, ckey) is not None:
            continue
        dense = _alloc_dense((n, k), dt, mod.qweight.device)
        if dense is None:
            break
        _lib.check(L.gbx_dequantize(mod.qweight.data_ptr(), mod.scales.data_ptr(), mod.zeros.data_ptr(), dense.data_ptr(), n, k, mod.bits,
                                    mod.group_size, _DT[dt], _DT[dt], _stream_ptr(dense.device)))
        _dense_cache.put(src, ckey, dense)
    # the interleaved gate | up image of every SwiGLU MLP (built by one long call on a dummy activation)
    for mod in model.modules():
        gate, up = getattr(mod, "gate_proj", None), getattr(mod, "up_proj", None)
        if (type(gate) is QuantizedLinear and type(up) is QuantizedLinear and gate.scales is not None and gate.qweight.is_cuda
                and 2 * gate.output_dims * gate.input_dims * 2 <= _dense_cache.budget - _dense_cache.nbytes):
            x = torch.zeros((1024, gate.input_dims), dtype=dtype or gate.scales.dtype, device=gate.qweight.device)
            quantized_swiglu(x, gate, up)
    return _dense_cache.nbytes


def dense_cache_bytes() -> int:
    """Bytes of HBM the resident images hold right now."""
    return _dense_cache.nbytes


def _alloc_dense(shape, dtype, device):
    """Storage for a new image: on out-of-memory the cache is emptied and the allocation retried once; None = give up (the caller
    falls back to the workspace path, which needs no resident image)."""
    try:
        return torch.empty(shape, dtype=dtype, device=device)
    except torch.cuda.OutOfMemoryError:
        _dense_cache.clear()
        torch.cuda.empty_cache()
        try:
            return torch.empty(shape, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            return None


def quantized_matmul(x: torch.Tensor, w: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor,
                     transpose: bool = True, group_size: int = 64, bits: int = 4,
                     bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[..., n] = sum_k x[..., k] * (scales[n, k//g] * q[n, k] + biases[n, k//g])  (+ bias[n]).

    ``w`` is the packed uint32 ``qweight [N, K*bits/32]``; ``biases`` are the GBA "zeros"
    (already MLX-signed, quantized_linear_gba.py:151-155).  Only ``transpose=True`` exists on
    this path.  Raises ValueError on shape/bits/group violations like the reference's asserts.
    """
    if not transpose:
        raise NotImplementedError("quantized_matmul(transpose=False) is not on the QuantizedLinear path")
    for t, nm in ((x, "x"), (w, "w"), (scales, "scales"), (biases, "biases")):
        _require_cuda(t, nm)
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"activations must be float16 or bfloat16, got {x.dtype}")
    if w.dtype not in (torch.uint32, torch.int32):
        raise ValueError(f"packed weight must be uint32, got {w.dtype}")
    if w.dim() != 2 or scales.dim() != 2 or biases.shape != scales.shape:
        raise ValueError("w must be [N, K*bits/32]; scales/biases must be [N, K/group_size]")
    n = w.shape[0]
    if (w.shape[1] * 32) % bits:
        raise ValueError(f"packed row of {w.shape[1]} words is not a whole number of {bits}-bit fields")
    k = w.shape[1] * 32 // bits
    if x.shape[-1] != k:
        raise ValueError(f"last dim of x ({x.shape[-1]}) != input_dims ({k}) implied by w and bits")
    if k % group_size or scales.shape != (n, k // group_size):
        raise ValueError(f"scales must be [{n}, {k}//{group_size}], got {tuple(scales.shape)}")
    if scales.dtype != x.dtype:
        scales = scales.to(x.dtype)
    if biases.dtype != x.dtype:
        biases = biases.to(x.dtype)
    if bias is not None:
        _require_cuda(bias, "bias")
        if bias.shape != (n,):
            raise ValueError(f"bias must be [{n}]")
        if bias.dtype != x.dtype:
            bias = bias.to(x.dtype)
        bias = bias.contiguous()
    x2 = x.reshape(-1, k).contiguous()
    m = x2.shape[0]
    src = (w, scales, biases)                           # the caller's tensor objects: identity of a resident dense image
    w = w.contiguous()
    scales = scales.contiguous()
    biases = biases.contiguous()
    if out is None:
        out = torch.empty((m, n), dtype=x.dtype, device=x.device)
    elif out.shape != (m, n) or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("out must be a contiguous [M, N] tensor of the activation dtype")
    L = _lib.lib()
    need = L.gbx_qlinear_workspace_bytes(m, n, k, bits, group_size) if (2 <= m <= 32 or m > 64) else 0
    dense_path = m >= 1024 and need >= n * k * 2 and 31 <= L.gbx_qlinear_kernel_id(m, n, k, bits, group_size, 1) <= 33
    cacheable = (dense_path and _dense_cache.budget >= n * k * 2 and m <= 61440 and all(a is b for a, b in zip((w, scales, biases), src))
                 and not torch.cuda.is_current_stream_capturing())
    ckey = (n, k, bits, group_size, x.dtype, x.device.index)
    if (not cacheable and 256 <= m < 4096 and k % 64 == 0 and _dense_cache.budget > 0 and all(a is b for a, b in zip((w, scales, biases), src))):
        # short prompt and the bf16 image of this weight is ALREADY resident (an earlier long call, or dense_cache_prewarm): the dense
        # GEMM beats the fused dequantising tiles from 256 rows on (Llama-3-8B layer at 512 rows: 398 -> 250 us); never built here
        dense = _dense_cache.get(src, ckey)
        if dense is not None:
            part = L.gbx_dense_workspace_bytes(m, n, k)
            scratch = torch.empty((part,), dtype=torch.uint8, device=x.device) if part else None
            rc = L.gbx_dense_forward_ws(x2.data_ptr(), dense.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        m, n, k, _DT[x.dtype], scratch.data_ptr() if part else None, part, _stream_ptr(x.device))
            if rc == _lib.GBX_OK:
                return out.reshape(*x.shape[:-1], n)
    if cacheable:
        # prompt-sized call on the dense path: the [N, K] image is built once (gbx_dequantize), kept resident, and every call
        # runs the GEMM alone (gbx_dense_forward_ws: k split over workgroups with fp32 scratch for few-tile shapes)
        dense = _dense_cache.get(src, ckey)
        if dense is None:
            dense = _alloc_dense((n, k), x.dtype, x.device)
            if dense is not None:
                _lib.check(L.gbx_dequantize(w.data_ptr(), scales.data_ptr(), biases.data_ptr(), dense.data_ptr(), n, k, bits, group_size,
                                            _DT[x.dtype], _DT[x.dtype], _stream_ptr(x.device)))
                _dense_cache.put(src, ckey, dense)
        if dense is not None:
            part = L.gbx_dense_workspace_bytes(m, n, k)
            scratch = torch.empty((part,), dtype=torch.uint8, device=x.device) if part else None
            rc = L.gbx_dense_forward_ws(x2.data_ptr(), dense.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        m, n, k, _DT[x.dtype], scratch.data_ptr() if part else None, part, _stream_ptr(x.device))
            if rc == _lib.GBX_OK:
                return out.reshape(*x.shape[:-1], n)
    if need:                                            # decode batches / short prompts: k split across workgroups; long prompts: dequantise once
        ws = torch.empty((need,), dtype=torch.uint8, device=x.device)      # caller-owned scratch (caching allocator: no sync)
        rc = L.gbx_qlinear_forward_ws(
            x2.data_ptr(), w.data_ptr(), scales.data_ptr(), biases.data_ptr(),
            bias.data_ptr() if bias is not None else None, out.data_ptr(),
            m, n, k, bits, group_size, _DT[x.dtype], ws.data_ptr(), need, _stream_ptr(x.device))
    else:
        rc = L.gbx_qlinear_forward(
            x2.data_ptr(), w.data_ptr(), scales.data_ptr(), biases.data_ptr(),
            bias.data_ptr() if bias is not None else None, out.data_ptr(),
            m, n, k, bits, group_size, _DT[x.dtype], _stream_ptr(x.device))
    _lib.check(rc)
    return out.reshape(*x.shape[:-1], n)


def unpack(w: torch.Tensor, bits: int) -> torch.Tensor:
    """Kernel-side unpack dump: packed uint32 [N, W] -> int32 [N, K] (bit-exact parity surface)."""
    _require_cuda(w, "w")
    n = w.shape[0]
    k = w.shape[1] * 32 // bits
    q = torch.empty((n, k), dtype=torch.int32, device=w.device)
    w = w.contiguous()
    _lib.check(_lib.lib().gbx_unpack(w.data_ptr(), q.data_ptr(), n, k, bits, _stream_ptr(w.device)))
    return q


def dequantize(w: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor, group_size: int = 64, bits: int = 4,
               dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``mx.dequantize`` role (gbx_lm/tuner/lora.py:62-68): scales*q + biases, shape [N, K]."""
    for t, nm in ((w, "w"), (scales, "scales"), (biases, "biases")):
        _require_cuda(t, nm)
    if scales.dtype not in (torch.float16, torch.bfloat16) or biases.dtype != scales.dtype:
        raise ValueError("scales/biases must both be float16 or bfloat16")
    dtype = dtype or scales.dtype
    n = w.shape[0]
    k = w.shape[1] * 32 // bits
    out = torch.empty((n, k), dtype=dtype, device=w.device)
    rc = _lib.lib().gbx_dequantize(w.contiguous().data_ptr(), scales.contiguous().data_ptr(),
                                   biases.contiguous().data_ptr(), out.data_ptr(), n, k, bits, group_size,
                                   _DT[scales.dtype], _DT[dtype], _stream_ptr(w.device))
    _lib.check(rc)
    return out


# ------------------------------------------------------------------------------------------------ fused decode block
def _companion(mod, kind: str, others, build):
    """Load-time companions (packed scale | zero tables, concatenated q|k|v) are functions of the modules' buffers only: built once per
    module and kept on it, so that a decode runner rebuilt for a grown KV cache reuses them (a rebuild cost 180 ms of re-packing before;
    scratch/gen_rate.py).  The key holds the buffers' addresses: replacing a buffer (a reload, a test that swaps scales) rebuilds."""
    mods = (mod,) + tuple(m for m in others if m is not None)
    key = (kind,) + tuple(t.data_ptr() for m in mods for t in (m.qweight, m.scales, m.zeros)) + tuple(m.scales.dtype for m in mods)
    memo = mod.__dict__.setdefault("_gbx_companions", {})
    hit = memo.get(kind)
    if hit is not None and hit[0] == key:
        return hit[1]
    out = build()
    memo[kind] = (key, out)
    return out


def pack_scales_zeros(mod, mod2=None) -> torch.Tensor:
    return _companion(mod, "sz2" if mod2 is not None else "sz1", (mod2,), lambda: _pack_scales_zeros(mod, mod2))


def _pack_scales_zeros(mod, mod2=None) -> torch.Tensor:
    """Load-time packed companion of ``(scales, zeros)`` for the decode GEMV (gbx_pack_scales_zeros): uint32
    ``[N/2, G, 2]`` for row pairs of one module, or ``[N, G, 2]`` interleaving ``mod`` / ``mod2`` (gate / up).
    A q/k/v launch takes ``torch.cat`` of its modules' packs."""
    n, g = mod.scales.shape
    for t in (mod.scales, mod.zeros) + ((mod2.scales, mod2.zeros) if mod2 is not None else ()):
        _require_cuda(t, "scales/zeros")
        if t.shape != (n, g) or t.dtype != mod.scales.dtype or not t.is_contiguous():
            raise ValueError("scales/zeros of the packed pair must be contiguous and agree in shape and dtype")
    out = torch.empty((n if mod2 is not None else n // 2, g, 2), dtype=torch.int32, device=mod.scales.device)
    _lib.check(_lib.lib().gbx_pack_scales_zeros(mod.scales.data_ptr(), mod.zeros.data_ptr(),
                                                mod2.scales.data_ptr() if mod2 is not None else None,
                                                mod2.zeros.data_ptr() if mod2 is not None else None,
                                                out.data_ptr(), n, g, _stream_ptr(out.device)))
    return out


def quantized_matmul_multi(x: torch.Tensor, mods, pre_norm=None) -> list:
    """[m(x) for m in mods] for 2-3 QuantizedLinear modules that share ``x`` (q/k/v, gate/up) -- ONE launch of the
    split-K matrix-core kernel when the batch is decode-sized (gbx_qlinear_forward_multi), else one call per module.
    ``pre_norm`` (an RMSNorm module): the modules see ``pre_norm(x)``; at one token the norm runs in the GEMV's prologue."""
    import ctypes
    m0 = mods[0]
    k = m0.input_dims
    x2 = x.reshape(-1, k)
    rows = x2.shape[0]
    same = all((m.bits, m.group_size, m.input_dims) == (m0.bits, m0.group_size, k) for m in mods)
    if pre_norm is not None:
        vpl = 64 if m0.bits == 2 else 32
        if (same and x.is_cuda and rows == 1 and 1 <= len(mods) <= 3 and k % 128 == 0 and k // vpl <= 128 and x.dtype in _DT
                and pre_norm.weight.dtype == x.dtype
                and all(m.bias is None and m.scales is not None and m.scales.dtype == x.dtype and m.zeros.dtype == x.dtype for m in mods)):
            x1 = x2.contiguous()
            ys = [torch.empty((1, m.output_dims), dtype=x.dtype, device=x.device) for m in mods]
            qlinear_fused(make_fused_desc(x1, mods, ys, norm_weight=pre_norm.weight, eps=pre_norm.eps), x.device)
            return [y.reshape(*x.shape[:-1], m.output_dims) for y, m in zip(ys, mods)]
        x = pre_norm(x)
        x2 = x.reshape(-1, k)
    if (same and x.is_cuda and rows == 1 and 1 < len(mods) <= 3 and k % 128 == 0 and x.dtype in _DT
            and all(m.bias is None and m.scales is not None and m.scales.dtype == x.dtype and m.zeros.dtype == x.dtype for m in mods)):
        # one token: the multi-segment decode GEMV (gbx_qlinear_fused_forward without norm / residual fusion)
        x1 = x2.contiguous()
        ys = [torch.empty((1, m.output_dims), dtype=x.dtype, device=x.device) for m in mods]
        try:
            qlinear_fused(make_fused_desc(x1, mods, ys), x.device)
        except NotImplementedError:                     # a shape the streaming kernel does not take: one call per module
            return [m(x) for m in mods]
        return [y.reshape(*x.shape[:-1], m.output_dims) for y, m in zip(ys, mods)]
    if not (same and x.is_cuda and 2 <= rows <= 64 and k >= 512 and k % 128 == 0 and 1 < len(mods) <= 3
            and all(m.output_dims % 32 == 0 and m.scales is not None and m.scales.dtype == x.dtype for m in mods)):
        return [m(x) for m in mods]
    x2 = x2.contiguous()
    ys = [torch.empty((rows, m.output_dims), dtype=x.dtype, device=x.device) for m in mods]
    n = len(mods)
    VP, I32 = ctypes.c_void_p * n, ctypes.c_int32 * n
    _lib.check(_lib.lib().gbx_qlinear_forward_multi(
        x2.data_ptr(), n, VP(*[m.qweight.data_ptr() for m in mods]), VP(*[m.scales.data_ptr() for m in mods]),
        VP(*[m.zeros.data_ptr() for m in mods]), VP(*[(m.bias.data_ptr() if m.bias is not None else None) for m in mods]),
        VP(*[y.data_ptr() for y in ys]), I32(*[m.output_dims for m in mods]), rows, k, m0.bits, m0.group_size,
        _DT[x.dtype], _stream_ptr(x.device)))
    return [y.reshape(*x.shape[:-1], m.output_dims) for y, m in zip(ys, mods)]


def gather_quantized_matmul(x: torch.Tensor, w: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor,
                            rhs_indices: torch.Tensor, transpose: bool = True, group_size: int = 64, bits: int = 4,
                            sorted_indices: bool = False, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mx.gather_qmm(x, w[E,N,W], scales[E,N,G], biases[E,N,G], rhs_indices=indices, transpose=True, ...)`` as
    QuantizedSwitchLinear calls it (gbx_lm/models/switch_layers.py:78-92), plus the optional ``bias[indices]``.

    x: [..., 1, 1, K] or [..., k, 1, K]; rhs_indices: [..., k] -> [..., k, 1, N].  Few (token, slot) pairs run the
    expert-indexed decode GEMV in ONE launch; many pairs are grouped by expert on the host (one batched GEMM per
    expert present -- the reference's ``sorted_indices`` hint needs no special handling)."""
    if not transpose:
        raise NotImplementedError("gather_quantized_matmul supports transpose=True only (the reference's only use)")
    for t, nm in ((x, "x"), (w, "w"), (scales, "scales"), (biases, "biases"), (rhs_indices, "rhs_indices")):
        _require_cuda(t, nm)
    E, N = w.shape[0], w.shape[1]
    K = x.shape[-1]
    if w.shape[2] * 32 != K * bits:
        raise ValueError(f"packed width {w.shape[2]} does not match K={K} at {bits} bits")
    if group_size not in (32, 64, 128):
        raise ValueError(f"The group size value ({group_size}) must be 32, 64 or 128.")
    k = rhs_indices.shape[-1]
    lead = tuple(rhs_indices.shape[:-1])
    if x.shape[-2] != 1 or x.shape[-3] not in (1, k) or tuple(x.shape[:-3]) != lead:
        raise ValueError(f"x {tuple(x.shape)} does not broadcast against indices {tuple(rhs_indices.shape)}")
    shared = x.shape[-3] == 1
    pairs = rhs_indices.numel()
    if x.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"x must be float16 or bfloat16 (got {x.dtype})")
    # scales / zeros / bias are read as the activation dtype by the kernels: cast here rather than reinterpret their bits
    # (the loader aligns them once, so this is a no-op on the decode path)
    if scales.dtype != x.dtype:
        scales = scales.to(x.dtype)
    if biases.dtype != x.dtype:
        biases = biases.to(x.dtype)
    if bias is not None and bias.dtype != x.dtype:
        bias = bias.to(x.dtype)
    w, scales, biases = w.contiguous(), scales.contiguous(), biases.contiguous()
    xr = x.reshape(-1, K).contiguous()                  # [tokens, K] (shared) or [pairs, K]
    idx = rhs_indices.reshape(-1)
    y = torch.empty((pairs, N), dtype=x.dtype, device=x.device)
    max_k = 512 * (64 if bits == 2 else 32)
    if pairs <= 64 and K % 128 == 0 and K <= max_k:
        idx32 = idx if idx.dtype == torch.int32 else idx.to(torch.int32)
        _lib.check(_lib.lib().gbx_gather_qlinear_forward(
            xr.data_ptr(), w.data_ptr(), scales.data_ptr(), biases.data_ptr(), bias.data_ptr() if bias is not None else None,
            idx32.data_ptr(), y.data_ptr(), pairs, k if shared else 1, E, N, K, bits, group_size, _DT[x.dtype],
            _stream_ptr(x.device)))
    else:                                               # grouped by expert (one host sync for the expert list)
        if torch.cuda.is_current_stream_capturing():
            raise NotCapturable(f"gather_quantized_matmul: {pairs} (token, expert) pairs / K={K} need the host-grouped path, which "
                               "cannot be captured into a HIP graph (the expert-indexed launch covers <= 64 pairs, K % 128 == 0)")
        ar = torch.arange(pairs, device=x.device)
        rows = ar // k if shared else ar
        for e in torch.unique(idx).tolist():
            sel = (idx == e).nonzero(as_tuple=True)[0]
            y[sel] = quantized_matmul(xr[rows[sel]], w[e], scales=scales[e], biases=biases[e], transpose=True,
                                      group_size=group_size, bits=bits, bias=bias[e] if bias is not None else None)
    return y.reshape(*lead, k, 1, N)


def gather_swiglu(x: torch.Tensor, gate, up, rhs_indices: torch.Tensor):
    """``silu(gate_proj(x, indices)) * up_proj(x, indices)`` of SwitchGLU for a decode-sized index set in ONE expert-indexed
    launch (gbx_gather_swiglu_forward); ``gate`` / ``up`` are QuantizedSwitchLinear modules.  Returns None when the shapes
    are outside that launch (the caller then runs the three ops)."""
    K, N, k = x.shape[-1], gate.output_dims, rhs_indices.shape[-1]
    pairs = rhs_indices.numel()
    max_k = 512 * (64 if gate.bits == 2 else 32)
    if not (x.is_cuda and x.dtype in _DT and pairs <= 64 and K % 128 == 0 and K <= max_k and x.shape[-2] == 1 and x.shape[-3] in (1, k)
            and (gate.bits, gate.group_size, gate.output_dims, gate.input_dims) == (up.bits, up.group_size, up.output_dims, up.input_dims)
            and getattr(gate, "bias", None) is None and getattr(up, "bias", None) is None
            and gate.scales.dtype == x.dtype and up.scales.dtype == x.dtype and rhs_indices.dtype == torch.int32):
        return None
    xr = x.reshape(-1, K).contiguous()
    idx = rhs_indices.reshape(-1).contiguous()
    y = torch.empty((pairs, N), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().gbx_gather_swiglu_forward(
        xr.data_ptr(), gate.qweight.data_ptr(), gate.scales.data_ptr(), gate.zeros.data_ptr(), up.qweight.data_ptr(),
        up.scales.data_ptr(), up.zeros.data_ptr(), idx.data_ptr(), y.data_ptr(), pairs, k if x.shape[-3] == 1 else 1,
        gate.qweight.shape[0], N, K, gate.bits, gate.group_size, _DT[x.dtype], _stream_ptr(x.device)))
    return y.reshape(*rhs_indices.shape[:-1], k, 1, N)


def quantized_swiglu(x: torch.Tensor, gate, up) -> Optional[torch.Tensor]:
    """``silu(gate_proj(x)) * up_proj(x)`` for a decode-sized batch (2..64 rows) in ONE launch (gbx_qlinear_swiglu_forward:
    the no-dequant matrix-core GEMM with the SwiGLU in its epilogue) or for a long prompt (the dense path of >= 1024 / 4096 rows:
    gbx_qlinear_swiglu_forward_ws).  Returns None when the shapes are outside both (the caller then runs the three ops)."""
    k, n = gate.input_dims, gate.output_dims
    x2 = x.reshape(-1, k)
    m = x2.shape[0]
    same = ((gate.bits, gate.group_size, gate.input_dims, gate.output_dims) == (up.bits, up.group_size, up.input_dims, up.output_dims)
            and gate.bias is None and up.bias is None and gate.scales is not None and up.scales is not None)
    if (x.is_cuda and x.dtype in _DT and 256 <= m < 1024 and same and gate.scales.dtype == x.dtype and up.scales.dtype == x.dtype
            and _dense_cache.budget > 0 and n % 256 == 0):
        # short prompt: only when the interleaved gate | up image is ALREADY resident (an earlier long call, dense_cache_prewarm)
        src = (gate.qweight, gate.scales, gate.zeros, up.qweight, up.scales, up.zeros)
        dense = _dense_cache.get(src, ("swiglu", n, k, gate.bits, gate.group_size, x.dtype, x.device.index))
        if dense is not None:
            x2c = x2.contiguous()
            y = torch.empty((m, n), dtype=x.dtype, device=x.device)
            rc = _lib.lib().gbx_dense_swiglu_forward(x2c.data_ptr(), dense.data_ptr(), y.data_ptr(), m, n, k, _DT[x.dtype], _stream_ptr(x.device))
            if rc == _lib.GBX_OK:
                return y.reshape(*x.shape[:-1], n)
    if x.is_cuda and x.dtype in _DT and m >= 1024 and same and gate.scales.dtype == x.dtype and up.scales.dtype == x.dtype:
        # long prompt: gate | up dequantised once into one dense matrix, one matrix-core GEMM with the SwiGLU in its epilogue
        need = _lib.lib().gbx_qlinear_swiglu_workspace_bytes(m, n, k)
        if need:
            x2 = x2.contiguous()
            y = torch.empty((m, n), dtype=x.dtype, device=x.device)
            src = (gate.qweight, gate.scales, gate.zeros, up.qweight, up.scales, up.zeros)
            ckey = ("swiglu", n, k, gate.bits, gate.group_size, x.dtype, x.device.index)
            dense = _dense_cache.get(src, ckey) if _dense_cache.budget > 0 else None
            if dense is not None:                       # the interleaved gate | up image of an earlier call: the GEMM alone
                rc = _lib.lib().gbx_dense_swiglu_forward(x2.data_ptr(), dense.data_ptr(), y.data_ptr(), m, n, k, _DT[x.dtype],
                                                         _stream_ptr(x.device))
                if rc == _lib.GBX_OK:
                    return y.reshape(*x.shape[:-1], n)
            ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
            _lib.check(_lib.lib().gbx_qlinear_swiglu_forward_ws(
                x2.data_ptr(), gate.qweight.data_ptr(), gate.scales.data_ptr(), gate.zeros.data_ptr(), up.qweight.data_ptr(),
                up.scales.data_ptr(), up.zeros.data_ptr(), y.data_ptr(), m, n, k, gate.bits, gate.group_size, _DT[x.dtype], ws.data_ptr(),
                need, _stream_ptr(x.device)))
            if all(t.is_contiguous() for t in src):
                _dense_cache.put(src, ckey, ws)
            return y.reshape(*x.shape[:-1], n)
    if not (x.is_cuda and x.dtype in _DT and 2 <= m <= 64 and k % 1024 == 0 and n % 16 == 0 and gate.bias is None and up.bias is None
            and (gate.bits, gate.group_size, gate.input_dims, gate.output_dims) == (up.bits, up.group_size, up.input_dims, up.output_dims)
            and (gate.group_size == 128 or gate.bits == 2) and gate.scales.dtype == x.dtype and up.scales.dtype == x.dtype):
        return None
    x2 = x2.contiguous()
    y = torch.empty((m, n), dtype=x.dtype, device=x.device)
    rc = _lib.lib().gbx_qlinear_swiglu_forward(x2.data_ptr(), gate.qweight.data_ptr(), gate.scales.data_ptr(), gate.zeros.data_ptr(),
                                               up.qweight.data_ptr(), up.scales.data_ptr(), up.zeros.data_ptr(), y.data_ptr(), m, n, k,
                                               gate.bits, gate.group_size, _DT[x.dtype], _stream_ptr(x.device))
    if rc == _lib.GBX_ERR_UNSUPPORTED:
        return None
    _lib.check(rc)
    return y.reshape(*x.shape[:-1], n)


def quantized_matmul_partials(x: torch.Tensor, mod):
    """The k-split batched-decode GEMM of ``mod`` WITHOUT its summing launch: ``(slices_ws fp32 [slices, M, N], slices)`` or
    None when this shape does not split k across workgroups (then call the module).  The consumer folds the sum in
    (``add_rmsnorm_slices``)."""
    import ctypes
    k, n = mod.input_dims, mod.output_dims
    x2 = x.reshape(-1, k)
    m = x2.shape[0]
    if not (x.is_cuda and x.dtype in _DT and 2 <= m <= 32 and mod.bias is None and mod.scales is not None and mod.scales.dtype == x.dtype):
        return None
    need = _lib.lib().gbx_qlinear_partials_bytes(m, n, k, mod.bits, mod.group_size)
    if not need:
        return None
    ws = torch.empty((need // 4,), dtype=torch.float32, device=x.device)
    slices = ctypes.c_int32(0)
    rc = _lib.lib().gbx_qlinear_forward_partials(x2.contiguous().data_ptr(), mod.qweight.data_ptr(), mod.scales.data_ptr(), mod.zeros.data_ptr(),
                                                 ws.data_ptr(), need, ctypes.byref(slices), m, n, k, mod.bits, mod.group_size,
                                                 _DT[x.dtype], _stream_ptr(x.device))
    if rc == _lib.GBX_ERR_UNSUPPORTED:
        return None
    _lib.check(rc)
    return ws, int(slices.value)


_XCD_LAYOUT = {}


def xcd_layout_ok(device) -> bool:
    """Does this device take the launches with XCD-local hand-offs (gbx_xcd_layout_check: 256 CUs in one partition, no CU mask, and a
    census launch that finds consecutive workgroups on 8 different XCDs)?  Checked once per device, outside any capture; False keeps
    every caller on the launch-per-op block."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return False
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _XCD_LAYOUT:
        L = _lib.lib()
        ok = False
        if hasattr(L, "gbx_xcd_layout_check") and not torch.cuda.is_current_stream_capturing():
            with torch.cuda.device(idx):
                ok = L.gbx_xcd_layout_check() == _lib.GBX_OK
            if not ok:
                import warnings
                warnings.warn("gbx_lm_amd: " + L.gbx_last_error().decode("utf-8", "replace") + " -- decoding on the launch-per-op block")
        elif torch.cuda.is_current_stream_capturing():
            return False                       # (not cached: ask again outside the capture)
        _XCD_LAYOUT[idx] = ok
    return _XCD_LAYOUT[idx]


def norm_state(device) -> torch.Tensor:
    """Zeroed state buffer of the chip-wide slice-sum + norm launches (gbx_norm_state_bytes): one per stream, created before any
    graph capture (its generation words must survive replays)."""
    return torch.zeros((_lib.lib().gbx_norm_state_bytes(),), dtype=torch.uint8, device=device)


NORM_STATUS_OFFSET = 16384      # include/gbx_qlinear.h: GBX_NORM_STATUS_OFFSET


def norm_status(state: torch.Tensor) -> int:
    """0, or 1 after a launch on ``state`` whose polls ran out (synchronises: outside the timed path)."""
    return int(state[NORM_STATUS_OFFSET:NORM_STATUS_OFFSET + 4].view(torch.int32).item())


def add_rmsnorm_slices(h: torch.Tensor, ws: torch.Tensor, slices: int, w: torch.Tensor, eps: float, state: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h += round(sum of the fp32 slices) in place, returns RMSNorm(h) * w -- one launch (gbx_add_rmsnorm_slices; with ``state`` and a
    shape it covers, the chip-wide form gbx_add_rmsnorm_slices_xcd)."""
    k = h.shape[-1]
    assert h.is_contiguous() and ws.numel() >= slices * h.numel()
    y = torch.empty_like(h)
    m = h.numel() // k
    if state is not None and env.flag("GBX_NORM_XCD") and k % 512 == 0 and k <= 4096 and m <= 8 * (32 // (k // 512)):
        _lib.check(_lib.lib().gbx_add_rmsnorm_slices_xcd(h.data_ptr(), ws.data_ptr(), slices, w.data_ptr(), y.data_ptr(), m, k, float(eps),
                                                         _DT[h.dtype], state.data_ptr(), _stream_ptr(h.device)))
        return y
    _lib.check(_lib.lib().gbx_add_rmsnorm_slices(h.data_ptr(), ws.data_ptr(), slices, w.data_ptr(), y.data_ptr(), m, k,
                                                 float(eps), _DT[h.dtype], _stream_ptr(h.device)))
    return y


def dense_linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ w.T (+ bias)`` for unquantised 16-bit weights ``w [N, K]`` on the hand-written matrix-core GEMM
    (gbx_dense_forward): the output head of a decode batch.  Falls back to ``F.linear`` for shapes outside it."""
    k, n = w.shape[1], w.shape[0]
    x2 = x.reshape(-1, k)
    if not (x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.dtype == x.dtype and w.is_contiguous() and k % 64 == 0
            and (bias is None or bias.dtype == x.dtype) and n * k < (1 << 30) and x2.shape[0] * k < (1 << 30)):
        return torch.nn.functional.linear(x, w, bias)
    x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().gbx_dense_forward(x2.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                            x2.shape[0], n, k, _DT[x.dtype], _stream_ptr(x.device)))
    return y.reshape(*x.shape[:-1], n)


def prepack_tiles(mod, mod_up=None):
    """Load-time tiled form of a 2-bit / group-128 QuantizedLinear for the matrix-core decode GEMV
    (gbx_qlinear_prepack_tiles): returns ``(wt, szt)`` int32 tensors.  With ``mod_up`` the pair (gate_proj, up_proj) is
    interleaved 8 + 8 rows per tile for a SwiGLU launch."""
    n, k = mod.output_dims, mod.input_dims
    mods = (mod,) if mod_up is None else (mod, mod_up)
    for m in mods:
        if (m.output_dims, m.input_dims, m.bits, m.group_size) != (n, k, mod.bits, mod.group_size):
            raise ValueError("the two matrices of a SwiGLU pair must agree in shape, bits and group_size")
        for t in (m.qweight, m.scales, m.zeros):
            _require_cuda(t, "QuantizedLinear tensor")
    rows = ((n + 15) // 16 * 16) if mod_up is None else ((n + 7) // 8 * 16)
    wt = torch.empty((rows * (k // 16),), dtype=torch.int32, device=mod.qweight.device)
    szt = torch.empty((rows * (k // 128),), dtype=torch.int32, device=mod.qweight.device)
    up = mod_up
    _lib.check_exp(_lib.exp_lib().gbx_qlinear_prepack_tiles(
        mod.qweight.data_ptr(), mod.scales.data_ptr(), mod.zeros.data_ptr(),
        up.qweight.data_ptr() if up is not None else None, up.scales.data_ptr() if up is not None else None,
        up.zeros.data_ptr() if up is not None else None,
        wt.data_ptr(), szt.data_ptr(), n, k, mod.bits, mod.group_size, _stream_ptr(wt.device)))
    return wt, szt


def gemv_tiled(x: torch.Tensor, wt: torch.Tensor, szt: torch.Tensor, n: int, k: int, out: Optional[torch.Tensor] = None,
               bits: int = 2, group_size: int = 128, norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0,
               residual: Optional[torch.Tensor] = None, swiglu: bool = False) -> torch.Tensor:
    """y[n] = W x from the tiled form (gbx_gemv_tiled); x is one activation row; optional fused RMSNorm prologue,
    residual epilogue or (pair tiles) SwiGLU epilogue."""
    _require_cuda(x, "x")
    if out is None:
        out = torch.empty((n,), dtype=x.dtype, device=x.device)
    _lib.check_exp(_lib.exp_lib().gbx_gemv_tiled(x.data_ptr(), wt.data_ptr(), szt.data_ptr(), out.data_ptr(), n, k, bits, group_size,
                                         norm_weight.data_ptr() if norm_weight is not None else None, float(eps),
                                         residual.data_ptr() if residual is not None else None, int(bool(swiglu)),
                                         _DT[x.dtype], _stream_ptr(x.device)))
    return out


def make_fused_desc(x: torch.Tensor, mods, ys, norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0,
                    residual: Optional[torch.Tensor] = None, swiglu: bool = False, packed=None,
                    bias_elsewhere: bool = False, x_acc: Optional[torch.Tensor] = None, x_out: Optional[torch.Tensor] = None,
                    zero_buf: Optional[torch.Tensor] = None) -> "_lib.FusedDesc":
    """Descriptor for one launch over 1-3 QuantizedLinear modules sharing ``x`` (q/k/v, gate/up, or a single one).
    ``packed``: ``pack_scales_zeros`` companion of the launch (per-module packs concatenated in order, or the one
    gate/up-interleaved tensor with ``swiglu``).
    ``x_acc`` (int64 [K], the fixed-point o_proj sums of ``attn_oproj_decode``): the launch forms x <- x + round(x_acc * 2^-32)
    in its prologue and writes it to ``x_out`` once; ``zero_buf`` (int64): cleared by this launch for a later one.
    All tensors must stay alive and in place while the descriptor is used (the decode runner owns static buffers)."""
    m0 = mods[0]
    for m in mods:
        if (m.bits, m.group_size, m.input_dims) != (m0.bits, m0.group_size, m0.input_dims):
            raise ValueError("fused segments must share bits, group_size and input_dims")
        if m.bias is not None and not bias_elsewhere:     # (q/k/v biases can ride in the attention launch instead)
            raise NotImplementedError("fused decode path does not carry a bias")
        for t in (m.qweight, m.scales, m.zeros):
            _require_cuda(t, "QuantizedLinear tensor")
        if m.scales.dtype != x.dtype or m.zeros.dtype != x.dtype:
            raise ValueError("scales/zeros must already be in the activation dtype")
    exp = x_acc is not None or x_out is not None or zero_buf is not None      # fields of the experimental library's descriptor
    d = _lib.ExpFusedDesc() if exp else _lib.FusedDesc()
    d.x = x.data_ptr()
    d.K, d.bits, d.group_size, d.act_dtype = m0.input_dims, m0.bits, m0.group_size, _DT[x.dtype]
    d.nseg = len(mods)
    for i, m in enumerate(mods):
        d.qweight[i], d.scales[i], d.zeros[i], d.N[i] = m.qweight.data_ptr(), m.scales.data_ptr(), m.zeros.data_ptr(), m.output_dims
    for i, y in enumerate(ys):
        d.y[i] = y.data_ptr()
    d.norm_weight = norm_weight.data_ptr() if norm_weight is not None else None
    d.norm_eps = float(eps)
    d.residual = residual.data_ptr() if residual is not None else None
    d.swiglu = 1 if swiglu else 0
    if packed is not None:
        _require_cuda(packed, "packed scales/zeros")
        rows = sum(m.output_dims for m in mods)
        if packed.dtype != torch.int32 or packed.numel() != rows * (m0.input_dims // m0.group_size):
            raise ValueError("packed scales/zeros do not match the launch (int32, one dword per row and group)")
        d.sz = packed.data_ptr()
    if x_acc is not None:
        if x_acc.dtype != torch.int64 or x_acc.numel() != m0.input_dims:
            raise ValueError("x_acc must be int64 [K]")
        d.x_acc = x_acc.data_ptr()
        d.x_out = x_out.data_ptr() if x_out is not None else None
    if zero_buf is not None:
        if zero_buf.dtype != torch.int64:
            raise ValueError("zero_buf must be int64")
        d.zero_buf, d.zero_n = zero_buf.data_ptr(), zero_buf.numel()
    return d


def _exp_kernels_requested() -> bool:
    """GBX_GEMV_MXD / GBX_GEMV_MX: route the fused launches through libgbx_experimental.so (its scaled-MFMA GEMVs)."""
    return env.flag("GBX_GEMV_MXD") or env.flag("GBX_GEMV_MX")


def _as_exp_desc(desc) -> "_lib.ExpFusedDesc":
    if isinstance(desc, _lib.ExpFusedDesc):
        return desc
    import ctypes
    e = _lib.ExpFusedDesc()
    ctypes.memmove(ctypes.addressof(e), ctypes.addressof(desc), ctypes.sizeof(_lib.FusedDesc))
    return e


def qlinear_fused(desc, device) -> None:
    import ctypes
    if isinstance(desc, _lib.ExpFusedDesc) or _exp_kernels_requested():
        e = _as_exp_desc(desc)
        _lib.check_exp(_lib.exp_lib().gbx_exp_qlinear_fused_forward(ctypes.byref(e), _stream_ptr(device)))
        return
    _lib.check(_lib.lib().gbx_qlinear_fused_forward(ctypes.byref(desc), _stream_ptr(device)))


def dgemv_plan(n: int, k: int, bits: int, group_size: int, epilogue: int, with_norm: bool = False):
    """(rows_per_batch, ssq_partials) of the decode-block GEMV launch for this shape -- ssq_partials = how many partial sums
    of y^2 the launch leaves in an ``ssq_out`` buffer, 0 if it cannot -- or None when dgemv does not cover the shape."""
    import ctypes
    r, wg = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = _lib.lib().gbx_dgemv_plan(n, k, bits, group_size, epilogue, 1 if with_norm else 0, ctypes.byref(r), ctypes.byref(wg))
    if rc == _lib.GBX_ERR_UNSUPPORTED:
        return None
    _lib.check(rc)
    return r.value, wg.value


def dgemv_pack(mod, rows_per_batch: int, mod_up=None) -> torch.Tensor:
    return _companion(mod, f"dg{rows_per_batch}{'u' if mod_up is not None else ''}", (mod_up,), lambda: _dgemv_pack(mod, rows_per_batch, mod_up))


def qkv_concat(q_proj, k_proj, v_proj):
    """q | k | v as ONE QuantizedLinear (rows concatenated once, kept on q_proj): a one-segment launch."""
    def build():
        from .models.quantized_linear_gba import QuantizedLinear
        mods = (q_proj, k_proj, v_proj)
        cat = QuantizedLinear.__new__(QuantizedLinear)
        torch.nn.Module.__init__(cat)
        cat.group_size, cat.bits, cat.double_group_size = q_proj.group_size, q_proj.bits, 32
        cat.input_dims, cat.output_dims, cat.bias = q_proj.input_dims, sum(m.output_dims for m in mods), None
        for nm in ("qweight", "scales", "zeros"):
            cat.register_buffer(nm, torch.cat([getattr(m, nm) for m in mods]).contiguous())
        return cat
    return _companion(q_proj, "qkvcat", (k_proj, v_proj), build)


def _dgemv_pack(mod, rows_per_batch: int, mod_up=None) -> torch.Tensor:
    """Load-time (scale | zero << 16) companion of one dgemv launch in batch order (gbx_dgemv_pack): int32
    ``[N / R, G, R]``, or ``[N / (R/2), G, R]`` with gate (``mod``) and up (``mod_up``) rows of a SwiGLU launch."""
    n, g = mod.scales.shape
    for t in (mod.scales, mod.zeros) + ((mod_up.scales, mod_up.zeros) if mod_up is not None else ()):
        _require_cuda(t, "scales/zeros")
        if t.shape != (n, g) or t.dtype != mod.scales.dtype or not t.is_contiguous():
            raise ValueError("scales/zeros must be contiguous and agree in shape and dtype")
    ro = rows_per_batch // 2 if mod_up is not None else rows_per_batch
    out = torch.empty((n // ro, g, rows_per_batch), dtype=torch.int32, device=mod.scales.device)
    _lib.check(_lib.lib().gbx_dgemv_pack(mod.scales.data_ptr(), mod.zeros.data_ptr(),
                                         mod_up.scales.data_ptr() if mod_up is not None else None,
                                         mod_up.zeros.data_ptr() if mod_up is not None else None,
                                         out.data_ptr(), n, g, rows_per_batch, _stream_ptr(out.device)))
    return out


def make_dgemv_desc(x: torch.Tensor, mod, y: torch.Tensor, packed: torch.Tensor, rows_per_batch: int, mod_up=None,
                    norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0, residual: Optional[torch.Tensor] = None,
                    ssq_in: Optional[torch.Tensor] = None, ssq_out: Optional[torch.Tensor] = None,
                    x_acc: Optional[torch.Tensor] = None, x_out: Optional[torch.Tensor] = None,
                    zero_buf: Optional[torch.Tensor] = None) -> "_lib.DgDesc":
    """Descriptor of one decode-block GEMV launch (gbx_dgemv_forward): ``mod`` alone (optionally ``+ residual``), or
    ``silu(mod(x)) * mod_up(x)``.  ``ssq_out`` / ``ssq_in``: float32 [GBX_DG_SSQ_SLOTS] buffers that carry the RMSNorm
    statistic from the launch that writes ``x`` to the launch that normalises it.  ``x_acc`` (int64 [K + 1], the fused attention
    launch's fixed-point o_proj sums) / ``x_out``: the prologue forms ``x + round(x_acc * 2^-32)`` and writes it to ``x_out``;
    ``zero_buf`` (int64): cleared by this launch.  Tensors must stay alive and in place."""
    import ctypes
    for m in (mod,) + ((mod_up,) if mod_up is not None else ()):
        if m.bias is not None:
            raise NotImplementedError("dgemv launches carry no bias")
        for t in (m.qweight, m.scales, m.zeros):
            _require_cuda(t, "QuantizedLinear tensor")
        if m.scales.dtype != x.dtype:
            raise ValueError("scales/zeros must already be in the activation dtype")
    if mod_up is not None and (mod_up.bits, mod_up.group_size, mod_up.input_dims, mod_up.output_dims) != (
            mod.bits, mod.group_size, mod.input_dims, mod.output_dims):
        raise ValueError("gate and up must be packed alike")
    for t in (ssq_in, ssq_out):
        if t is not None and (t.dtype != torch.float32 or t.numel() != _lib.GBX_DG_SSQ_SLOTS or not t.is_contiguous()):
            raise ValueError("ssq buffers are float32 [GBX_DG_SSQ_SLOTS]")
    d = _lib.DgDesc()
    d.struct_size = ctypes.sizeof(_lib.DgDesc)
    d.epilogue = _lib.GBX_DG_SWIGLU if mod_up is not None else (_lib.GBX_DG_RESIDUAL if residual is not None else _lib.GBX_DG_PLAIN)
    d.x, d.qweight, d.sz, d.y = x.data_ptr(), mod.qweight.data_ptr(), packed.data_ptr(), y.data_ptr()
    d.qweight_up = mod_up.qweight.data_ptr() if mod_up is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    d.norm_weight = norm_weight.data_ptr() if norm_weight is not None else None
    d.ssq_in = ssq_in.data_ptr() if ssq_in is not None else None
    d.ssq_out = ssq_out.data_ptr() if ssq_out is not None else None
    d.N, d.K, d.bits, d.group_size, d.act_dtype = mod.output_dims, mod.input_dims, mod.bits, mod.group_size, _DT[x.dtype]
    d.rows_per_batch = rows_per_batch
    d.norm_eps = float(eps)
    if x_acc is not None:
        if x_acc.dtype != torch.int64 or x_acc.numel() != mod.input_dims + 1 or not x_acc.is_contiguous():
            raise ValueError("x_acc is int64 [K + 1]")
        d.x_acc = x_acc.data_ptr()
        d.x_out = x_out.data_ptr() if x_out is not None else None
    if zero_buf is not None:
        if zero_buf.dtype != torch.int64 or not zero_buf.is_contiguous():
            raise ValueError("zero_buf is a contiguous int64 tensor")
        d.zero_buf, d.zero_n = zero_buf.data_ptr(), zero_buf.numel()
    if packed.dtype != torch.int32 or packed.numel() != (2 if mod_up is not None else 1) * mod.output_dims * (mod.input_dims // mod.group_size):
        raise ValueError("packed scales/zeros do not match the launch")
    return d


def dgemv(desc, device) -> None:
    import ctypes
    _lib.check(_lib.lib().gbx_dgemv_forward(ctypes.byref(desc), _stream_ptr(device)))


def dgemv_kernel_name(desc) -> str:
    """Name of the kernel gbx_dgemv_forward launches for this descriptor, as rocprofv3 prints its template head (read back from
    the library, gbx_dgemv_kernel_id: bench labels and profile filters never re-derive the dispatch)."""
    import ctypes
    k, nw, r = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(_lib.lib().gbx_dgemv_kernel_id(ctypes.byref(desc), ctypes.byref(k), ctypes.byref(nw), ctypes.byref(r)))
    return f"{'dgemv_mx_kernel' if k.value == 1 else 'dgemv_kernel'} (NW {nw.value}, R {r.value})"


def dattn_supported(d_model: int, n_heads: int, n_kv: int, head_dim: int, qkv_mod, o_mod) -> bool:
    """Does the fused attention launch (gbx_dattn_forward) cover this block?"""
    L = _lib.lib()
    if not hasattr(L, "gbx_dattn_plan"):
        return False
    return L.gbx_dattn_plan(d_model, n_heads, n_kv, head_dim, qkv_mod.bits, qkv_mod.group_size, o_mod.bits, o_mod.group_size) == _lib.GBX_OK


def dattn_state(device) -> torch.Tensor:
    """Hand-off state of the fused attention launch: zero-filled once, private to one stream (one decode runner)."""
    return torch.zeros((_lib.lib().gbx_dattn_state_bytes(),), dtype=torch.uint8, device=device)


def make_dattn_desc(x: torch.Tensor, norm_weight: torch.Tensor, eps: float, qkv_mod, qkv_packed: torch.Tensor, kcache: torch.Tensor,
                    vcache: torch.Tensor, inv_freq: torch.Tensor, pos: torch.Tensor, o_mod, acc: torch.Tensor, state: torch.Tensor,
                    n_heads: int, capacity: int, rope_scale: float, sm_scale: float, traditional: bool,
                    ssq_in: Optional[torch.Tensor] = None, attn_out: Optional[torch.Tensor] = None,
                    x_acc: Optional[torch.Tensor] = None, x_out: Optional[torch.Tensor] = None,
                    zero_buf: Optional[torch.Tensor] = None) -> "_lib.DattnDesc":
    """Descriptor of one fused attention launch: RMSNorm -> q|k|v (``qkv_mod``: the three projections' rows concatenated) -> RoPE +
    cache append -> attention -> o_proj partial sums into ``acc`` (int64 [D + 1], 2^-32 fixed point)."""
    import ctypes
    D = x.numel()
    for t in (x, norm_weight, qkv_mod.qweight, qkv_packed, kcache, vcache, inv_freq, pos, o_mod.qweight, o_mod.scales, o_mod.zeros, acc, state):
        _require_cuda(t, "dattn tensor")
    if qkv_mod.bias is not None or o_mod.bias is not None:
        raise NotImplementedError("the fused attention launch carries no bias")
    if acc.dtype != torch.int64 or acc.numel() != D + 1:
        raise ValueError("acc is int64 [D + 1]")
    if qkv_mod.output_dims != (n_heads + 16) * 128 or qkv_mod.input_dims != D or o_mod.output_dims != D or o_mod.input_dims != n_heads * 128:
        raise ValueError("shapes do not describe a block with 8 kv heads of 128")
    if kcache.shape[0] != 1 or kcache.shape[1] != 8 or kcache.shape[2] != capacity or kcache.shape[3] != 128 or not kcache.is_contiguous():
        raise ValueError("kcache / vcache are [1, 8, capacity, 128]")
    if qkv_packed.dtype != torch.int32 or qkv_packed.numel() != qkv_mod.output_dims * (D // qkv_mod.group_size):
        raise ValueError("qkv_packed is gbx_dgemv_pack of the concatenated matrix, rows_per_batch 2")
    d = _lib.DattnDesc()
    d.struct_size = ctypes.sizeof(_lib.DattnDesc)
    d.act_dtype = _DT[x.dtype]
    d.x, d.norm_weight = x.data_ptr(), norm_weight.data_ptr()
    d.ssq_in = ssq_in.data_ptr() if ssq_in is not None else None
    d.qkv_qweight, d.qkv_sz = qkv_mod.qweight.data_ptr(), qkv_packed.data_ptr()
    d.kcache, d.vcache, d.inv_freq, d.pos = kcache.data_ptr(), vcache.data_ptr(), inv_freq.data_ptr(), pos.data_ptr()
    d.o_qweight, d.o_scales, d.o_zeros = o_mod.qweight.data_ptr(), o_mod.scales.data_ptr(), o_mod.zeros.data_ptr()
    d.acc, d.state = acc.data_ptr(), state.data_ptr()
    d.attn_out = attn_out.data_ptr() if attn_out is not None else None
    if x_acc is not None:               # the previous block's fused MLP launch left down_proj's output as fixed-point sums
        if x_acc.dtype != torch.int64 or x_acc.numel() != D + 1 or not x_acc.is_contiguous() or x_out is None or ssq_in is not None:
            raise ValueError("x_acc is int64 [D + 1], comes with x_out and without ssq_in")
        d.x_acc, d.x_out = x_acc.data_ptr(), x_out.data_ptr()
    if zero_buf is not None:
        if zero_buf.dtype != torch.int64 or not zero_buf.is_contiguous():
            raise ValueError("zero_buf is a contiguous int64 tensor")
        d.zero_buf, d.zero_n = zero_buf.data_ptr(), zero_buf.numel()
    d.D, d.n_heads, d.capacity = D, n_heads, capacity
    d.qkv_bits, d.qkv_group_size, d.o_bits, d.o_group_size = qkv_mod.bits, qkv_mod.group_size, o_mod.bits, o_mod.group_size
    d.traditional = 1 if traditional else 0
    d.norm_eps, d.rope_scale, d.sm_scale = float(eps), float(rope_scale), float(sm_scale)
    return d


def dmlp_supported(d_model: int, gate, up, down, dtype) -> bool:
    """Does the fused MLP launch (gbx_dmlp_forward) cover this block?"""
    L = _lib.lib()
    if not hasattr(L, "gbx_dmlp_plan") or any(m.bias is not None for m in (gate, up, down)):
        return False
    if (gate.bits, gate.group_size) != (up.bits, up.group_size) or gate.output_dims != up.output_dims or dtype not in _DT:
        return False
    return L.gbx_dmlp_plan(d_model, gate.output_dims, gate.bits, gate.group_size, down.bits, down.group_size, _DT[dtype]) == _lib.GBX_OK


def repack_planes(mod) -> torch.Tensor:
    """The 3-bit ``qweight`` of ``mod`` as a 2-bit and a 1-bit plane per 32 values (gbx_repack_planes): same shape and bytes, q = lo + 4 hi -- what the
    matrix-core fused MLP launch reads.  A load-time companion kept on the module; ``mod.qweight`` stays the loader's tensor."""
    def build():
        qw = mod.qweight
        _require_cuda(qw, "qweight")
        if mod.bits != 3 or not qw.is_contiguous():
            raise ValueError("repack_planes: contiguous 3-bit rows")
        out = torch.empty_like(qw)
        _lib.check(_lib.lib().gbx_repack_planes(qw.data_ptr(), out.data_ptr(), qw.shape[0], mod.input_dims, 3, _stream_ptr(qw.device)))
        return out
    return _companion(mod, "planes3", (), build)


def dmlp_weight_layout(d_model: int, gate, down, dtype) -> int:
    """GBX_LAYOUT_REFERENCE / GBX_LAYOUT_PLANES: the layout gbx_dmlp_forward wants for these modules (negative: unsupported)."""
    return _lib.lib().gbx_dmlp_weight_layout(d_model, gate.output_dims, gate.bits, gate.group_size, down.bits, down.group_size, _DT[dtype])


def make_dmlp_desc(x: torch.Tensor, x_acc: torch.Tensor, x_out: torch.Tensor, norm_weight: torch.Tensor, eps: float, gate, up, down,
                   gate_up_packed: torch.Tensor, down_packed: torch.Tensor, acc_out: torch.Tensor, state: torch.Tensor,
                   zero_buf: Optional[torch.Tensor] = None, act_out: Optional[torch.Tensor] = None) -> "_lib.DmlpDesc":
    """Descriptor of one fused MLP launch: x' = x + round(x_acc * 2^-32) (written to ``x_out``) -> RMSNorm -> gate | up + SwiGLU ->
    down_proj partial sums into ``acc_out`` (int64 [D + 1]).  ``gate_up_packed`` / ``down_packed``: dgemv_pack(gate, 2, up) / dgemv_pack(down, 2)."""
    import ctypes
    D, I = x.numel(), gate.output_dims
    for t in (x, x_acc, x_out, norm_weight, gate.qweight, up.qweight, down.qweight, gate_up_packed, down_packed, acc_out, state):
        _require_cuda(t, "dmlp tensor")
    if x_acc.dtype != torch.int64 or acc_out.dtype != torch.int64 or x_acc.numel() != D + 1 or acc_out.numel() != D + 1:
        raise ValueError("x_acc / acc_out are int64 [D + 1]")
    if down.output_dims != D or down.input_dims != I or gate.input_dims != D:
        raise ValueError("shapes do not describe an MLP block")
    if gate_up_packed.numel() != 2 * I * (D // gate.group_size) or down_packed.numel() != D * (I // down.group_size):
        raise ValueError("packed companions do not match (rows_per_batch 2)")
    d = _lib.DmlpDesc()
    d.struct_size = ctypes.sizeof(_lib.DmlpDesc)
    d.act_dtype = _DT[x.dtype]
    d.x, d.x_acc, d.x_out, d.norm_weight = x.data_ptr(), x_acc.data_ptr(), x_out.data_ptr(), norm_weight.data_ptr()
    # 3-bit rows: the matrix-core launch reads them as a 2-bit and a 1-bit plane (companions kept on the modules, built once)
    layout = dmlp_weight_layout(D, gate, down, x.dtype)
    if layout < 0:
        raise ValueError("gbx_dmlp_forward does not cover these shapes (dmlp_supported)")
    planes = layout == _lib.GBX_LAYOUT_PLANES
    gq, uq, dq = (repack_planes(gate), repack_planes(up), repack_planes(down)) if planes else (gate.qweight, up.qweight, down.qweight)
    d.weight_layout, d.reserved0 = layout, 0
    d._planes = (gq, uq, dq)                                     # (kept alive with the descriptor)
    d.gate_qweight, d.up_qweight, d.gate_up_sz = gq.data_ptr(), uq.data_ptr(), gate_up_packed.data_ptr()
    d.down_qweight, d.down_sz = dq.data_ptr(), down_packed.data_ptr()
    d.acc_out, d.state = acc_out.data_ptr(), state.data_ptr()
    if zero_buf is not None:
        if zero_buf.dtype != torch.int64 or not zero_buf.is_contiguous():
            raise ValueError("zero_buf is a contiguous int64 tensor")
        d.zero_buf, d.zero_n = zero_buf.data_ptr(), zero_buf.numel()
    d.act_out = act_out.data_ptr() if act_out is not None else None
    d.D, d.I, d.norm_eps = D, I, float(eps)
    d.gate_up_bits, d.gate_up_group_size, d.down_bits, d.down_group_size = gate.bits, gate.group_size, down.bits, down.group_size
    return d


def dmlp(desc, device) -> None:
    import ctypes
    _lib.check(_lib.lib().gbx_dmlp_forward(ctypes.byref(desc), _stream_ptr(device)))


def dblock_supported(d_model: int, n_heads: int, n_kv: int, head_dim: int, qkv_mod, o_mod, gate, up, down, dtype) -> bool:
    """Does the one-launch block (gbx_dblock_forward) cover this decoder block?"""
    L = _lib.lib()
    if not hasattr(L, "gbx_dblock_plan") or dtype not in _DT or any(m.bias is not None for m in (qkv_mod, o_mod, gate, up, down)):
        return False
    if (gate.bits, gate.group_size) != (up.bits, up.group_size) or gate.output_dims != up.output_dims:
        return False
    return L.gbx_dblock_plan(d_model, gate.output_dims, n_heads, n_kv, head_dim, qkv_mod.bits, qkv_mod.group_size, o_mod.bits, o_mod.group_size,
                             gate.bits, gate.group_size, down.bits, down.group_size, _DT[dtype]) == _lib.GBX_OK


def dblock_state(device) -> torch.Tensor:
    """Hand-off state of the one-launch block: zero-filled once, private to one stream (one decode runner)."""
    return torch.zeros((_lib.lib().gbx_dblock_state_bytes(),), dtype=torch.uint8, device=device)


def dblock_max_context() -> int:
    """Cached positions up to which the one-launch block sweeps the context itself (beyond: the two-launch block's cooperative sweep)."""
    return int(_lib.lib().gbx_dblock_max_context())


def dblock_status_word(state: torch.Tensor) -> torch.Tensor:
    """The status word of a gbx_dblock state buffer as a one-element int32 device view (hand area of 8 x 4096 granules, then word 8 * 16)."""
    off = 8 * 4096 * 8 + 8 * 16 * 4
    return state[off:off + 4].view(torch.int32)


def make_dblock_desc(d_a: "_lib.DattnDesc", d_m: "_lib.DmlpDesc", state: torch.Tensor, keep=()) -> "_lib.DblockDesc":
    """The one-launch form of a (fused attention, fused MLP) descriptor pair: same tensors, same accumulators, same clears -- the
    attention launch's zero_buf and the MLP launch's zero_buf both become this launch's (gbx_dblock)."""
    import ctypes
    _require_cuda(state, "dblock state")
    if d_m.x_acc != d_a.acc:
        raise ValueError("the MLP launch must consume the attention launch's accumulator")
    d = _lib.DblockDesc()
    d.struct_size = ctypes.sizeof(_lib.DblockDesc)
    d.act_dtype = d_a.act_dtype
    for f in ("x", "norm_weight", "ssq_in", "qkv_qweight", "qkv_sz", "kcache", "vcache", "inv_freq", "pos", "o_qweight", "o_scales", "o_zeros",
              "x_acc", "x_out", "attn_out", "capacity", "traditional", "norm_eps", "rope_scale", "sm_scale"):
        setattr(d, f, getattr(d_a, f))
    d.acc_o = d_a.acc
    d.norm2_weight, d.norm2_eps = d_m.norm_weight, d_m.norm_eps
    for f in ("gate_qweight", "up_qweight", "gate_up_sz", "down_qweight", "down_sz", "act_out"):
        setattr(d, f, getattr(d_m, f))
    d.acc_d, d.x2_out = d_m.acc_out, d_m.x_out
    d.zero_buf0, d.zero_n0 = d_a.zero_buf, d_a.zero_n
    d.zero_buf1, d.zero_n1 = d_m.zero_buf, d_m.zero_n
    d.state = state.data_ptr()
    d._keep = (d_a, d_m, state) + tuple(keep)
    return d


def dblock(desc, device) -> None:
    import ctypes
    _lib.check(_lib.lib().gbx_dblock_forward(ctypes.byref(desc), _stream_ptr(device)))


def dblock_status(state: torch.Tensor) -> int:
    import ctypes
    c = ctypes.c_int32(0)
    _lib.check(_lib.lib().gbx_dblock_status(state.data_ptr(), ctypes.byref(c)))
    return c.value


def acc_residual(x: torch.Tensor, acc: torch.Tensor, out: torch.Tensor, zero=()) -> None:
    """out <- round(x + round(acc * 2^-32)); ``acc`` and up to three more [D + 1] accumulators are cleared (gbx_acc_residual)."""
    z = [t.data_ptr() for t in zero] + [None] * (3 - len(zero))
    _lib.check(_lib.lib().gbx_acc_residual(x.data_ptr(), acc.data_ptr(), out.data_ptr(), x.numel(), _DT[x.dtype], z[0], z[1], z[2], _stream_ptr(x.device)))


def dattn(desc, device) -> None:
    import ctypes
    _lib.check(_lib.lib().gbx_dattn_forward(ctypes.byref(desc), _stream_ptr(device)))


DATTN_STATUS_BYTE_OFFSET = 8 * 2048 * 8 + 8 * 16 * 4     # dattn_impl.h: DA_GROUPS * DA_HAND_STRIDE granules, then state word 8 * 16


def dattn_status_word(state: torch.Tensor) -> torch.Tensor:
    """The launch status word inside a fused launch's state buffer as a one-element int32 view (device memory: copy it to the host
    together with the token, gbx_lm_amd.utils.generate_step does)."""
    return state[DATTN_STATUS_BYTE_OFFSET:DATTN_STATUS_BYTE_OFFSET + 4].view(torch.int32)


def dattn_status(state: torch.Tensor) -> int:
    """0, or the hand-off of a fused attention launch that timed out (synchronises; call off the timed path)."""
    import ctypes
    c = ctypes.c_int32(0)
    _lib.check(_lib.lib().gbx_dattn_status(state.data_ptr(), ctypes.byref(c)))
    return c.value


class StreamProgram:
    """Device-resident program of the streamed decode GEMV / persistent engine (gbx_stream_encode / gbx_stream_run):
    one entry per ``make_fused_desc`` descriptor.  ``run(begin, end)`` enqueues ONE launch that walks ops [begin, end)
    with the packed rows of later ops streaming into LDS while earlier ones are consumed."""

    def __init__(self, descs, device, dtype):
        import ctypes
        n = len(descs)
        arr = (_lib.ExpFusedDesc * n)(*[_as_exp_desc(d) for d in descs])
        host = (ctypes.c_uint8 * (n * _lib.GBX_STREAM_OP_BYTES))()
        _lib.check_exp(_lib.exp_lib().gbx_stream_encode(arr, n, ctypes.addressof(host), len(host)))
        self.n = n
        self.dtype = dtype
        self.device = torch.device(device)
        self._host = host                  # op `begin` of a launch rides in the kernel arguments
        self.blob = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(self.device)
        self._keep = descs                 # the descriptors' tensors are owned by the caller (decode runner)

    def run(self, begin: int = 0, end: Optional[int] = None) -> None:
        end = self.n if end is None else end
        import ctypes
        _lib.check_exp(_lib.exp_lib().gbx_stream_run(self.blob.data_ptr(), ctypes.addressof(self._host), begin, end, _DT[self.dtype],
                                             _stream_ptr(self.device)))


def attn_decode(q, k, v, kcache, vcache, inv_freq, pos, out, n_heads: int, n_kv: int, head_dim: int, capacity: int,
                rope_scale: float, sm_scale: float, traditional: bool) -> None:
    _lib.check(_lib.lib().gbx_attn_decode(q.data_ptr(), k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                          inv_freq.data_ptr(), pos.data_ptr(), out.data_ptr(), n_heads, n_kv, head_dim,
                                          capacity, float(rope_scale), float(sm_scale), int(bool(traditional)),
                                          _DT[q.dtype], _stream_ptr(q.device)))


ATTN_OPROJ_HANDOFF_WORDS = 288     # GBX_ATTN_OPROJ_HANDOFF_WORDS (include/gbx_experimental.h)


def attn_oproj_supported(n_heads: int, n_kv: int, head_dim: int, o_proj) -> bool:
    return bool(_lib.exp_lib().gbx_attn_oproj_supported(n_heads, n_kv, head_dim, o_proj.output_dims, o_proj.bits, o_proj.group_size)) \
        and o_proj.input_dims == n_heads * head_dim and o_proj.bias is None


def attn_oproj_decode(q, k, v, kcache, vcache, inv_freq, pos, o_proj, acc, n_heads: int, n_kv: int, head_dim: int, capacity: int,
                      rope_scale: float, sm_scale: float, traditional: bool) -> None:
    """RoPE + cache append + attention + o_proj of one new token (batch 1, short cache) in one launch; the o_proj sums are
    ADDED into ``acc[:D]`` (int64, fixed point 2^-32); ``acc`` is int64 [D + n_kv * ATTN_OPROJ_HANDOFF_WORDS], all zero on
    entry (the tail is the in-launch hand-off area) -- gbx_attn_oproj_decode."""
    if acc.dtype != torch.int64 or acc.numel() != o_proj.output_dims + n_kv * ATTN_OPROJ_HANDOFF_WORDS:
        raise ValueError("acc must be int64 [D + n_kv * ATTN_OPROJ_HANDOFF_WORDS]")
    _lib.check_exp(_lib.exp_lib().gbx_attn_oproj_decode(q.data_ptr(), k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                                inv_freq.data_ptr(), pos.data_ptr(), o_proj.qweight.data_ptr(),
                                                o_proj.scales.data_ptr(), o_proj.zeros.data_ptr(), acc.data_ptr(), n_heads, n_kv,
                                                head_dim, capacity, o_proj.output_dims, o_proj.bits, o_proj.group_size,
                                                float(rope_scale), float(sm_scale), int(bool(traditional)), _DT[q.dtype],
                                                _stream_ptr(q.device)))


def attn_decode_batched(q, k, v, kcache, vcache, inv_freq, pos, out, n_heads: int, n_kv: int, head_dim: int,
                        rope_scale: float, sm_scale: float, traditional: bool, q_norm_w=None, k_norm_w=None,
                        norm_eps: float = 0.0, q_bias=None, k_bias=None, v_bias=None, workspace=None) -> None:
    """One new token per sequence, all at position ``pos`` (device int64): q/out [B, n_heads*hd], k/v [B, n_kv*hd],
    caches [B, n_kv, capacity, hd] (gbx_attn_decode_batched: RoPE + cache append + softmax(q k^T) v in one launch)."""
    _require_cuda(q, "q")
    b, cap = kcache.shape[0], kcache.shape[2]
    if workspace is None:                               # long caches: scratch for the context split (flash-decoding)
        need = _lib.lib().gbx_attn_workspace_bytes(b, n_heads, n_kv, head_dim, cap)
        workspace = torch.empty((need,), dtype=torch.uint8, device=q.device) if need else None
    _lib.check(_lib.lib().gbx_attn_decode_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                             inv_freq.data_ptr(), pos.data_ptr(), out.data_ptr(), b, n_heads, n_kv, head_dim,
                                                 cap, float(rope_scale), float(sm_scale), int(bool(traditional)),
                                                 q_norm_w.data_ptr() if q_norm_w is not None else None,
                                                 k_norm_w.data_ptr() if k_norm_w is not None else None, float(norm_eps),
                                                 q_bias.data_ptr() if q_bias is not None else None,
                                                 k_bias.data_ptr() if k_bias is not None else None,
                                                 v_bias.data_ptr() if v_bias is not None else None,
                                                 _DT[q.dtype], workspace.data_ptr() if workspace is not None else None,
                                                 workspace.numel() if workspace is not None else 0, _stream_ptr(q.device)))


def attn_decode_slices(qkv_ws: torch.Tensor, slices: int, kcache, vcache, inv_freq, pos, out, n_heads: int, n_kv: int, head_dim: int,
                       rope_scale: float, sm_scale: float, traditional: bool, q_norm_w=None, k_norm_w=None, norm_eps: float = 0.0,
                       workspace=None) -> None:
    """``attn_decode_batched`` whose q | k | v are the fp32 slices ``qkv_ws [slices, B, (n_heads + 2 n_kv) * hd]`` that
    ``quantized_matmul_partials`` left for the row-concatenated projection (gbx_attn_decode_slices): the sum of the slices is formed,
    rounded like the GEMM's own summing launch, inside the attention launch."""
    b, cap = kcache.shape[0], kcache.shape[2]
    assert qkv_ws.dtype == torch.float32 and qkv_ws.numel() >= slices * b * (n_heads + 2 * n_kv) * head_dim
    if workspace is None:
        need = _lib.lib().gbx_attn_workspace_bytes(b, n_heads, n_kv, head_dim, cap)
        workspace = torch.empty((need,), dtype=torch.uint8, device=out.device) if need else None
    _lib.check(_lib.lib().gbx_attn_decode_slices(qkv_ws.data_ptr(), int(slices), kcache.data_ptr(), vcache.data_ptr(), inv_freq.data_ptr(),
                                                 pos.data_ptr(), out.data_ptr(), b, n_heads, n_kv, head_dim, cap, float(rope_scale),
                                                 float(sm_scale), int(bool(traditional)),
                                                 q_norm_w.data_ptr() if q_norm_w is not None else None,
                                                 k_norm_w.data_ptr() if k_norm_w is not None else None, float(norm_eps), _DT[out.dtype],
                                                 workspace.data_ptr() if workspace is not None else None,
                                                 workspace.numel() if workspace is not None else 0, _stream_ptr(out.device)))


def attn_cached(q: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, pos: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(q K^T * scale) V for ONE query token per sequence over cache positions 0..pos (device int64), the new
    token's k / v already stored: q [B, Hq, 1, D] (rotated), caches [B, Hkv, cap, D] -> [B, Hq, 1, D] (gbx_attn_cached)."""
    _require_cuda(q, "q")
    b, hq, _, d = q.shape
    qc = q.contiguous()
    out = torch.empty_like(qc)
    _lib.check(_lib.lib().gbx_attn_cached(qc.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), pos.data_ptr(), out.data_ptr(), b, hq,
                                          kcache.shape[1], d, kcache.shape[2], float(scale), _DT[q.dtype], _stream_ptr(q.device)))
    return out


def attn_prefill_supported(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    """shapes / layouts gbx_attn_prefill takes without a copy: head_dim 64 / 128 contiguous, key rows head_dim apart"""
    d = q.shape[3]
    return (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and k.dtype == q.dtype and v.dtype == q.dtype and d in (64, 128)
            and k.shape[3] == d and v.shape[3] == d and q.shape[1] % k.shape[1] == 0 and k.shape[1] == v.shape[1]
            and q.stride(3) == 1 and k.stride(3) == 1 and v.stride(3) == 1 and k.stride(2) == d and v.stride(2) == d
            and all(t.stride(i) % 8 == 0 for t in (q, k, v) for i in (0, 1, 2)) and all(t.data_ptr() % 16 == 0 for t in (q, k, v)))


def attn_prefill(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, offset: int, scale: float) -> torch.Tensor:
    """Causal attention of a prompt chunk (gbx_attn_prefill): q [B, Hq, L, D] at positions offset .. offset + L - 1, k / v
    [B, Hkv, S, D] with S >= offset + L (any batch / head strides) -> context [B, L, Hq * D] (what o_proj consumes)."""
    _require_cuda(q, "q")
    b, hq, l, d = q.shape
    out = torch.empty((b, l, hq * d), dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().gbx_attn_prefill(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), b, hq, k.shape[1], l, k.shape[2], d,
                                           q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                           int(offset), float(scale), _DT[q.dtype], _stream_ptr(q.device)))
    return out


def moe_route(logits: torch.Tensor, top_k: int, mode: int = 0, renormalize: bool = True):
    """Router logits [..., E] -> (indices int32 [..., k], scores [..., k]) in one launch (gbx_moe_route)."""
    _require_cuda(logits, "logits")
    e = logits.shape[-1]
    lg = logits.reshape(-1, e).contiguous()
    inds = torch.empty((lg.shape[0], top_k), dtype=torch.int32, device=logits.device)
    scores = torch.empty((lg.shape[0], top_k), dtype=logits.dtype, device=logits.device)
    _lib.check(_lib.lib().gbx_moe_route(lg.data_ptr(), inds.data_ptr(), scores.data_ptr(), lg.shape[0], e, top_k, mode,
                                        int(bool(renormalize)), _DT[logits.dtype], _stream_ptr(logits.device)))
    lead = tuple(logits.shape[:-1])
    return inds.reshape(*lead, top_k), scores.reshape(*lead, top_k)


def moe_combine(y: torch.Tensor, scores: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_s scores[..., s] * y[..., s, :] (+ residual, the decoder layer's ``h + mlp(...)``) in one launch
    (gbx_moe_combine_residual)."""
    _require_cuda(y, "y")
    k, d = y.shape[-2], y.shape[-1]
    yc, sc = y.contiguous(), scores.contiguous()
    out = torch.empty(tuple(y.shape[:-2]) + (d,), dtype=y.dtype, device=y.device)
    rc = residual.contiguous() if residual is not None else None
    _lib.check(_lib.lib().gbx_moe_combine_residual(yc.data_ptr(), sc.data_ptr(), rc.data_ptr() if rc is not None else None,
                                                   out.data_ptr(), yc.numel() // (k * d), k, d, _DT[y.dtype], _stream_ptr(y.device)))
    return out


def qlinear_residual(mod, x: torch.Tensor, residual: torch.Tensor) -> Optional[torch.Tensor]:
    """``residual + mod(x)`` for ONE token through the decode GEMV with the add in its epilogue (gbx_qlinear_fused_forward);
    None when the call is outside that launch (several rows, bias, unsupported K) -- the caller then adds itself."""
    k = mod.input_dims
    vpl = 64 if mod.bits == 2 else 32
    if not (x.is_cuda and x.numel() == k and residual.numel() == mod.output_dims and mod.bias is None and k % 128 == 0
            and k // vpl <= 512 and x.dtype in _DT and mod.scales is not None and mod.scales.dtype == x.dtype
            and mod.zeros.dtype == x.dtype and residual.dtype == x.dtype):
        return None
    x1, r1 = x.reshape(1, k).contiguous(), residual.reshape(1, -1).contiguous()
    y = torch.empty_like(r1)
    qlinear_fused(make_fused_desc(x1, (mod,), (y,), residual=r1), x.device)
    return y.reshape(residual.shape)


def kv_append(kcache: torch.Tensor, vcache: torch.Tensor, k: torch.Tensor, v: torch.Tensor, pos: torch.Tensor) -> None:
    """caches[:, :, pos] = k, v for one new token per sequence at the device-resident position -- one launch
    (gbx_kv_append).  caches [B, H, cap, Dk|Dv] contiguous, k / v [B, H, 1, Dk|Dv]."""
    _require_cuda(k, "k")
    b, h, cap, dk = kcache.shape
    kc, vc = k.contiguous(), v.contiguous()
    _lib.check(_lib.lib().gbx_kv_append(kc.data_ptr(), vc.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), pos.data_ptr(),
                                        b * h, cap, dk, vcache.shape[3], _DT[k.dtype], _stream_ptr(k.device)))


def logprobs_argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None, token: Optional[torch.Tensor] = None,
                    workspace: Optional[torch.Tensor] = None, status=(), packed: Optional[torch.Tensor] = None):
    """(logits - logsumexp(logits), argmax) for one row of logits [1, V] or [V] on the device (gbx_logprobs_argmax):
    returns (logprobs with the shape of ``logits``, token int64 [1]).  ``out`` / ``token`` / ``workspace``: caller-owned buffers (a
    captured graph).  ``packed`` (int64 [4]) with up to three one-element int32 ``status`` views: packed = (token, status words) --
    what generate_step sends to the host after a step, in one copy (gbx_logprobs_argmax_packed)."""
    _require_cuda(logits, "logits")
    lg = logits.contiguous()
    v = lg.shape[-1]
    assert lg.numel() == v, "one row of logits"
    L = _lib.lib()
    out = torch.empty_like(lg) if out is None else out
    token = torch.empty((1,), dtype=torch.int64, device=lg.device) if token is None else token
    ws = torch.empty((L.gbx_logprobs_workspace_bytes(),), dtype=torch.uint8, device=lg.device) if workspace is None else workspace
    assert out.is_contiguous() and out.numel() == v and out.dtype == lg.dtype and token.dtype == torch.int64
    if packed is None:
        _lib.check(L.gbx_logprobs_argmax(lg.data_ptr(), out.data_ptr(), token.data_ptr(), ws.data_ptr(), v, _DT[lg.dtype],
                                         _stream_ptr(lg.device)))
    else:
        assert packed.dtype == torch.int64 and packed.numel() >= 4 and len(status) <= 3
        assert all(w.dtype == torch.int32 and w.numel() == 1 for w in status)
        st = [w.data_ptr() for w in status] + [None] * (3 - len(status))
        _lib.check(L.gbx_logprobs_argmax_packed(lg.data_ptr(), out.data_ptr(), token.data_ptr(), ws.data_ptr(), v, _DT[lg.dtype],
                                                st[0], st[1], st[2], packed.data_ptr(), _stream_ptr(lg.device)))
    return out, token


def add_rmsnorm(h: torch.Tensor, delta: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """h += delta in place (activation-dtype add), returns RMSNorm(h) * w -- one launch (gbx_add_rmsnorm)."""
    _require_cuda(h, "h")
    k = h.shape[-1]
    assert h.is_contiguous() and delta.is_contiguous() and delta.shape == h.shape
    y = torch.empty_like(h)
    _lib.check(_lib.lib().gbx_add_rmsnorm(h.data_ptr(), delta.data_ptr(), w.data_ptr(), y.data_ptr(), h.numel() // k, k,
                                          float(eps), _DT[h.dtype], _stream_ptr(h.device)))
    return y


def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """silu(gate) * up with the op-by-op roundings of the model graph, one launch (gbx_swiglu)."""
    _require_cuda(gate, "gate")
    assert gate.is_contiguous() and up.is_contiguous() and gate.shape == up.shape
    y = torch.empty_like(gate)
    _lib.check(_lib.lib().gbx_swiglu(gate.data_ptr(), up.data_ptr(), y.data_ptr(), gate.numel(), _DT[gate.dtype],
                                     _stream_ptr(gate.device)))
    return y


def embed_next_token(keys: torch.Tensor, token_io: torch.Tensor, pos: torch.Tensor, table: torch.Tensor, h: torch.Tensor,
                     from_keys: bool, advance_pos: bool) -> None:
    """Device-side start of a decode step (see gbx_embed_next_token)."""
    _lib.check(_lib.lib().gbx_embed_next_token(keys.data_ptr(), token_io.data_ptr(), pos.data_ptr(), table.data_ptr(),
                                               h.data_ptr(), table.shape[1], table.shape[0], int(from_keys), int(advance_pos),
                                               _DT[h.dtype], _stream_ptr(h.device)))


def dense_gemv(x: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None,
               norm_weight: Optional[torch.Tensor] = None, eps: float = 0.0,
               argmax_keys: Optional[torch.Tensor] = None, index_offset: int = 0) -> torch.Tensor:
    """logits[N] = W[N,K] @ (x or RMSNorm(x)) for the unquantised head (qllama.py:194-198).  ``index_offset``: global index of
    row 0 when ``w`` is one rank's shard of the vocabulary (the argmax keys then carry global token ids)."""
    for t, nm in ((x, "x"), (w, "w")):
        _require_cuda(t, nm)
    n, k = w.shape
    if out is None:
        out = torch.empty((n,), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().gbx_dense_gemv_argmax_shard(x.data_ptr(), w.data_ptr(), out.data_ptr(),
                                                      norm_weight.data_ptr() if norm_weight is not None else None, float(eps),
                                                      n, k, argmax_keys.data_ptr() if argmax_keys is not None else None,
                                                      int(index_offset), _DT[x.dtype], _stream_ptr(x.device)))
    return out


def embed_next_token_sharded(gathered_keys: torch.Tensor, local_keys: torch.Tensor, token_io: torch.Tensor, pos: torch.Tensor,
                             table: torch.Tensor, h: torch.Tensor, advance_pos: bool) -> None:
    """Greedy token of a vocabulary-sharded head from every rank's argmax keys (gbx_embed_next_token_sharded)."""
    _lib.check(_lib.lib().gbx_embed_next_token_sharded(gathered_keys.data_ptr(), gathered_keys.numel(), local_keys.data_ptr(),
                                                       token_io.data_ptr(), pos.data_ptr(), table.data_ptr(), h.data_ptr(),
                                                       table.shape[1], table.shape[0], int(advance_pos), _DT[h.dtype],
                                                       _stream_ptr(h.device)))


def eminf_optimize(logits: torch.Tensor, alpha: float = 0.65, num_steps: Optional[int] = None, threshold: float = 0.05):
    """EMINF logits optimiser on the device (gbx_eminf_optimize): returns ``(best_logits fp32 [V], entropies fp32 [2])``
    with ``entropies = (best, initial)``; no host synchronisation."""
    _require_cuda(logits, "logits")
    x = logits.reshape(-1).contiguous()
    if x.dtype not in _DT and x.dtype != torch.float32:
        x = x.float()
    code = 2 if x.dtype == torch.float32 else _DT[x.dtype]
    v = x.numel()
    best = torch.empty((v,), dtype=torch.float32, device=x.device)
    scratch = torch.empty((v,), dtype=torch.float32, device=x.device)
    ent = torch.empty((2,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gbx_eminf_optimize(x.data_ptr(), code, best.data_ptr(), scratch.data_ptr(), ent.data_ptr(), v,
                                             float(alpha), float(threshold), -1 if num_steps is None else int(num_steps),
                                             _stream_ptr(x.device)))
    return best, ent


def rope(x: torch.Tensor, inv_freq: torch.Tensor, offset, scale: float = 1.0, traditional: bool = False,
         mscale: float = 1.0) -> torch.Tensor:
    """Rotary embedding of x [B, H, L, D] (any strides with a unit feature stride, e.g. the transposed projection output)
    in one launch (gbx_rope); ``offset`` is an int or a device int64 tensor.  Returns a contiguous tensor."""
    _require_cuda(x, "x")
    if x.dim() != 4 or x.stride(-1) != 1:
        x = x.reshape(-1, 1, x.shape[-2], x.shape[-1]).contiguous()
    B, H, L, D = x.shape
    y = torch.empty((B, H, L, D), dtype=x.dtype, device=x.device)
    pos_ptr, off = (offset.data_ptr(), 0) if isinstance(offset, torch.Tensor) else (None, int(offset))
    _lib.check(_lib.lib().gbx_rope(x.data_ptr(), y.data_ptr(), inv_freq.data_ptr(), pos_ptr, off, B, H, L, D,
                                   inv_freq.numel() * 2, x.stride(0), x.stride(1), x.stride(2), float(scale), float(mscale),
                                   int(bool(traditional)), _DT[x.dtype], _stream_ptr(x.device)))
    return y


def rope_kv_write(k: torch.Tensor, v: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, inv_freq: torch.Tensor,
                  pos0: int, t0: int, scale: float = 1.0, traditional: bool = False) -> None:
    """RoPE(k) of a prompt chunk written into rows t0.. of the key cache and v copied into the value cache, one launch
    (gbx_rope_kv_write).  k / v: [B, H, L, D] views with a unit feature stride; caches [B, H, capacity, D] contiguous."""
    _require_cuda(k, "k")
    B, H, L, D = k.shape
    _lib.check(_lib.lib().gbx_rope_kv_write(k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(), inv_freq.data_ptr(),
                                            int(pos0), B, H, L, D, k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1),
                                            v.stride(2), kcache.shape[2], int(t0), float(scale), int(bool(traditional)), _DT[k.dtype],
                                            _stream_ptr(k.device)))


def rope_append(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor,
                inv_freq: torch.Tensor, pos: torch.Tensor, n_heads: int, n_kv: int, head_dim: int,
                rope_scale: float, traditional: bool) -> None:
    """Batched one-token RoPE + KV-cache append (gbx_rope_append): q [B, n_heads*hd] rotated in place, k/v [B, n_kv*hd]
    written at the device-resident position ``pos`` of caches [B, n_kv, capacity, hd]."""
    for t in (q, k, v, kcache, vcache):
        _require_cuda(t, "rope_append tensor")
        if not t.is_contiguous():
            raise ValueError("rope_append needs contiguous tensors")
    _lib.check(_lib.lib().gbx_rope_append(q.data_ptr(), k.data_ptr(), v.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
                                          inv_freq.data_ptr(), pos.data_ptr(), q.shape[0], n_heads, n_kv, head_dim,
                                          kcache.shape[2], float(rope_scale), int(bool(traditional)), _DT[q.dtype],
                                          _stream_ptr(q.device)))


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_cuda(x, "x")
    k = x.shape[-1]
    x2 = x.reshape(-1, k).contiguous()
    if out is None:
        out = torch.empty_like(x2)
    _lib.check(_lib.lib().gbx_rmsnorm(x2.data_ptr(), w.data_ptr(), out.data_ptr(), x2.shape[0], k, float(eps),
                                      _DT[x.dtype], _stream_ptr(x.device)))
    return out.reshape(x.shape)
