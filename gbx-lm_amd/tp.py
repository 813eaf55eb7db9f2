"""Tensor-parallel sharding of the QuantizedLinear layers over the GPUs of one node (RCCL over xGMI).

Build-side extension: the reference has no tensor parallelism.  The packed format shards naturally:

  * q/k/v/gate/up  -- split OUTPUT rows N (whole packed rows + their scales/zeros move together): no communication.
  * o_proj/down    -- consume those sharded activations, so split INPUT columns K at multiples of 128 values
                      (a whole number of 32-value periods, groups and 16-byte slots for every bits/group in scope);
                      each rank produces a full-N partial sum -> ONE all-reduce(sum) after each.
  * lm_head        -- (round 3) split over the VOCABULARY: rank r streams rows [r V/world, (r+1) V/world) of the 16-bit head
                      (at TP = 8 a replicated head was 1.05 GB per rank per token next to 0.27 GB of sharded QuantizedLinear:
                      a <= 2.4x ceiling).  Greedy decode exchanges 64 argmax keys per rank (one 512-byte all-gather); callers
                      that need the logits get them all-gathered.

One process per GPU; ``torch.distributed`` backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .models.quantized_linear_gba import QuantizedLinear


def shard_rows(m: QuantizedLinear, rank: int, world: int) -> None:
    """Keep output rows [rank*N/world, (rank+1)*N/world)."""
    n = m.output_dims
    if n % world:
        raise ValueError(f"output_dims {n} not divisible by TP degree {world}")
    lo, hi = rank * n // world, (rank + 1) * n // world
    for nm in ("qweight", "scales", "zeros"):
        m._buffers[nm] = m._buffers[nm][lo:hi].contiguous()
    if m.bias is not None:
        m._buffers["bias"] = m._buffers["bias"][lo:hi].contiguous()
    m.output_dims = hi - lo


def shard_cols(m: QuantizedLinear, rank: int, world: int) -> None:
    """Keep input columns [rank*K/world, (rank+1)*K/world); bias stays on rank 0 only (added once by the sum)."""
    k = m.input_dims
    if k % world or (k // world) % 128:
        raise ValueError(f"input_dims {k} / TP {world} must be a multiple of 128 values")
    kl = k // world
    w_lo, w_hi = rank * kl * m.bits // 32, (rank + 1) * kl * m.bits // 32
    g_lo, g_hi = rank * kl // m.group_size, (rank + 1) * kl // m.group_size
    m._buffers["qweight"] = m._buffers["qweight"][:, w_lo:w_hi].contiguous()
    m._buffers["scales"] = m._buffers["scales"][:, g_lo:g_hi].contiguous()
    m._buffers["zeros"] = m._buffers["zeros"][:, g_lo:g_hi].contiguous()
    m._buffers["channel_scale"] = m._buffers["channel_scale"][..., rank * kl:(rank + 1) * kl].contiguous()
    if m.bias is not None and rank != 0:
        m._buffers["bias"] = torch.zeros_like(m._buffers["bias"])
    m.input_dims = kl


def shard_llama(model, rank: int, world: int, group: Optional[dist.ProcessGroup] = None):
    """In-place TP sharding of a qllama.Model; attention heads and KV heads are split with their projections."""
    if world == 1:
        return model
    for blk in model.model.layers:
        at, mlp = blk.self_attn, blk.mlp
        if at.n_heads % world or at.n_kv_heads % world:
            raise ValueError(f"heads {at.n_heads}/{at.n_kv_heads} not divisible by TP degree {world}")
        for p in (at.q_proj, at.k_proj, at.v_proj, mlp.gate_proj, mlp.up_proj):
            shard_rows(p, rank, world)
        for p in (at.o_proj, mlp.down_proj):
            shard_cols(p, rank, world)
        at.n_heads //= world
        at.n_kv_heads //= world
        at.tp_group, mlp.tp_group = group, group
        at.tp_world, mlp.tp_world = world, world
    model.tp_world, model.tp_rank, model.tp_group = world, rank, group
    hw = model.head_weight()
    v = hw.shape[0]
    # vocabulary shard of the output head (the embedding lookup keeps the table) -- only for model classes whose head_weight() /
    # forward consult the shard (qllama.Model); others keep the full head and `tp_vocab` unset, so no runner applies an offset
    code = getattr(getattr(type(model), "head_weight", None), "__code__", None)
    honours = code is not None and ("_tp_head" in code.co_consts or "_tp_head" in code.co_names)
    if honours and v % world == 0 and (v // world) % 2 == 0:
        lo, hi = rank * v // world, (rank + 1) * v // world
        model._tp_head = hw[lo:hi].detach().clone().contiguous()
        model.tp_vocab = (lo, hi, v)
        if not model.args.tie_word_embeddings:          # an untied head is not needed in full any more
            model.lm_head.weight = torch.nn.Parameter(model._tp_head, requires_grad=False)
    return model


_flat_gather = {}          # process group -> does its backend implement all_gather_into_tensor (decided ONCE, never by exception)


def _has_flat_gather(group) -> bool:
    key = id(group) if group is not None else None
    if key not in _flat_gather:
        # gloo (the CPU tests) has no flat form; nccl (= RCCL) has.  A capability of the backend, so every rank answers alike --
        # an exception-driven fallback around a collective would let ONE rank enqueue a different collective than its peers.
        _flat_gather[key] = str(dist.get_backend(group)).lower() != "gloo"
    return _flat_gather[key]


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None) -> torch.Tensor:
    """out[r * n : (r + 1) * n] = rank r's ``inp`` (flat, n = inp.numel()); one collective, the same one on every rank."""
    if _has_flat_gather(group):
        dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), group=group)
    else:
        parts = [torch.empty_like(inp.reshape(-1)) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, inp.reshape(-1), group=group)
        out.view(-1).copy_(torch.cat(parts))
    return out


def gather_vocab(logits_local: torch.Tensor, model) -> torch.Tensor:
    """[..., V / world] shards of the logits -> [..., V] on every rank (rank order = vocabulary order)."""
    world = model.tp_world
    flat = logits_local.reshape(-1, logits_local.shape[-1]).contiguous()
    out = torch.empty((world,) + tuple(flat.shape), dtype=flat.dtype, device=flat.device)
    all_gather_into(out, flat, getattr(model, "tp_group", None))
    return out.permute(1, 0, 2).reshape(*logits_local.shape[:-1], -1)


class P2PUnavailable(RuntimeError):
    """Some rank could not set up its peer-mapped inbox (no fine-grained device memory)."""


class P2PAllReduce:
    """One-shot push all-reduce of a short 16-bit vector through peer-mapped inboxes (gbx_p2p_allreduce, csrc/p2p_allreduce.hip):
    every rank writes its vector into every peer's memory over xGMI and sums what arrived in its own, in rank order, fp32, one
    rounding -- identical bits on every rank, one launch, graph-capturable.  The inbox handles travel through ``group`` once at
    construction (any backend).  Validated on ranks that share one GPU (tests/test_p2p_gpu.py); opt-in for real multi-GPU runs
    (``GBX_TP_P2P=1``, see ``enable_p2p``) until a multi-GPU box has confirmed it."""

    def __init__(self, rank: int, world: int, max_elems: int, device, group=None):
        import ctypes
        from . import _lib
        self.L = _lib.lib()
        self.rank, self.world, self.max_elems = rank, world, int(max_elems)
        self.device = torch.device(device)
        self.mapped, self.own = [], None
        with torch.cuda.device(self.device):
            own = ctypes.c_void_p()
            handle = ctypes.create_string_buffer(64)
            reason = None
            try:          # fine-grained memory or nothing (gbx_p2p_inbox_create has no coarse-grained fallback)
                _lib.check(self.L.gbx_p2p_inbox_create(world, self.max_elems, ctypes.byref(own), handle))
                self.own = own
            except (NotImplementedError, RuntimeError) as e:
                reason = f"rank {rank}: {e}"
            # every rank learns every rank's outcome BEFORE anyone maps or pushes: one failing rank switches all of them back
            outcomes = [None] * world
            dist.all_gather_object(outcomes, (reason, bytes(handle.raw)), group=group)
            bad = [r for r, _ in outcomes if r is not None]
            if bad:
                self.close()
                raise P2PUnavailable("; ".join(bad))
            arr = (ctypes.c_void_p * world)()
            for p in range(world):
                if p == rank:
                    arr[p] = own.value
                    continue
                m = ctypes.c_void_p()
                _lib.check(self.L.gbx_p2p_inbox_open(ctypes.create_string_buffer(outcomes[p][1], 64), ctypes.byref(m)))
                self.mapped.append(m)
                arr[p] = m.value
            self.inboxes = arr
        dist.barrier(group=group)                       # nobody pushes before everybody has mapped

    def check(self) -> None:
        """Raise if a launch on this inbox gave up waiting for a peer (synchronous 4-byte read: call it off the timed path)."""
        import ctypes
        from . import _lib
        flag = ctypes.c_int32(0)
        _lib.check(self.L.gbx_p2p_status(self.own, self.world, self.max_elems, ctypes.byref(flag)))
        if flag.value:
            raise RuntimeError(f"P2PAllReduce (rank {self.rank}): a one-shot all-reduce timed out waiting for a peer; its result is invalid")

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        from . import _lib
        from .ops import _DT, _stream_ptr
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float16) and t.numel() % 2 == 0
                and t.numel() <= self.max_elems):
            raise ValueError("P2PAllReduce: contiguous bf16 / fp16 CUDA tensor with an even element count <= max_elems")
        _lib.check(self.L.gbx_p2p_allreduce(t.data_ptr(), t.numel(), self.rank, self.world, self.inboxes, self.max_elems,
                                            _DT[t.dtype], _stream_ptr(t.device)))
        return t

    def close(self) -> None:
        for m in getattr(self, "mapped", []):
            self.L.gbx_p2p_inbox_close(m)
        self.mapped = []
        if getattr(self, "own", None) is not None:
            self.L.gbx_p2p_inbox_destroy(self.own)
            self.own = None


_p2p: Optional[P2PAllReduce] = None
p2p_unavailable_reason: Optional[str] = None      # why enable_p2p fell back to RCCL (reported in bench.py's config.tp_allreduce)


def enable_p2p(rank: int, world: int, max_elems: int, device, group=None) -> Optional[P2PAllReduce]:
    """Route ``all_reduce_`` of 16-bit tensors with at most ``max_elems`` elements through the one-shot P2P kernel.  Returns None --
    and leaves the RCCL all-reduce in place on EVERY rank, with the reason in ``p2p_unavailable_reason`` -- when any rank cannot
    provide a fine-grained inbox."""
    global _p2p, p2p_unavailable_reason
    try:
        _p2p = P2PAllReduce(rank, world, max_elems, device, group)
        p2p_unavailable_reason = None
    except P2PUnavailable as e:
        _p2p, p2p_unavailable_reason = None, str(e)
        import warnings
        warnings.warn(f"one-shot P2P all-reduce unavailable, keeping RCCL: {e}")
    return _p2p


def p2p_check() -> str:
    """Raise if a one-shot P2P all-reduce of this process gave up waiting for a peer (its result, and everything computed from it,
    is invalid); returns a short status for reports.  Synchronous 4-byte read: call it at synchronisation points, off the timed path."""
    if _p2p is None:
        return "p2p off"
    _p2p.check()
    return "p2p ok (no time-out)"


def all_reduce_(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of the per-rank partial outputs of a K-split layer (the path's only exchange step)."""
    if (_p2p is not None and t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.is_contiguous()
            and t.numel() % 2 == 0 and t.numel() <= _p2p.max_elems):
        return _p2p(t)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
