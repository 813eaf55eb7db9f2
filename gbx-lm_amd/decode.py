"""Batch-1 decode runner: the whole per-token launch sequence captured once in a HIP graph and replayed.

Counterpart of the reference's decode hot loop (gbx_lm/utils.py:325-338: ``_step`` -> ``model(y[None], cache)`` ->
last-position logits), restructured for MI355X: ~230 tiny launches per token are launch-bound from Python, so the
sequence is recorded once per KV-cache capacity with static buffers and a device-resident position, then replayed
(graph replay costs ~10-16 us per token instead of ~3-4 us per launch).  Numerics follow the eager graph in
``models/qllama.py`` op for op; tests compare the two.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F

from . import env, ops
from .models.cache import KVCache
from .tp import all_reduce_


class DecodeRunner:
    """Owns the captured graph for one model and one KV-cache capacity: one new token per sequence and replay.  The
    batch is whatever the KV cache holds (B = 1 for the reference's generate loop; B > 1 = synchronous batched decode,
    every sequence at the same position, QuantizedLinear then runs the split-K MFMA kernel)."""

    def __init__(self, model, caches: List[KVCache], use_graph: bool = True):
        self.model = model
        self.args = model.args
        self.dev = next(model.parameters()).device
        self.caches = caches
        self.capacity = caches[0].capacity
        self.dtype = model.model.embed_tokens.weight.dtype
        self.batch = caches[0].keys.shape[0] if caches[0].keys is not None else 1
        self.tok = torch.zeros((self.batch,), dtype=torch.long, device=self.dev)
        if getattr(self, "pos", None) is None:                   # (FusedDecodeRunner creates it first: its descriptors hold the address)
            self.pos = torch.zeros((1,), dtype=torch.long, device=self.dev)
        self.logits = None
        self.graph = None
        a0 = model.model.layers[0].self_attn
        self.n_heads, self.n_kv, self.hd = a0.n_heads, a0.n_kv_heads, a0.head_dim
        self.ar = torch.arange(self.capacity, device=self.dev)
        self.tp_world = getattr(model, "tp_world", 1)
        # generation words of the chip-wide slice-sum + norm launches (batched decode): created before any capture
        self.norm_state = ops.norm_state(self.dev) if (self.batch > 1 and self.dev.type == "cuda" and ops.xcd_layout_ok(self.dev)) else None
        # B > 1: q|k|v as ONE row-concatenated matrix (+7 MB per 2-bit 8B layer) so that the batch GEMM can split k across workgroups --
        # every workgroup then stages a quarter of the activations, the term that bounds it -- and leave fp32 slices that the
        # attention launch sums (gbx_attn_decode_slices); 11.5 -> 9.4 us per layer at B = 32
        self.qkv_cat = {}
        self._qkv_probe = {}
        if self.batch > 1 and self.tp_world == 1 and env.flag("GBX_QKV_SLICES"):
            from .models.quantized_linear_gba import QuantizedLinear
            for li, layer in enumerate(model.model.layers):
                at = layer.self_attn
                mods = (at.q_proj, at.k_proj, at.v_proj)
                if (len({(m.bits, m.group_size, m.input_dims) for m in mods}) == 1 and all(m.bias is None for m in mods)
                        and all(getattr(m, "scales", None) is not None and m.scales.dtype == self.dtype for m in mods)
                        and self.hd in (64, 128) and at.rope.inv_freq.numel() * 2 == self.hd
                        and type(at.rope).__name__ in ("RoPE", "Llama3RoPE")):
                    cat = QuantizedLinear(at.q_proj.input_dims, sum(m.output_dims for m in mods), False, at.q_proj.group_size, at.q_proj.bits)
                    for nm in ("qweight", "scales", "zeros"):
                        cat._buffers[nm] = torch.cat([getattr(m, nm) for m in mods]).contiguous()
                    # probe once per (bits, group, K, N) whether the k-split partials GEMM takes this batch and leaves few enough
                    # slices for the attention launch to sum (<= 8): otherwise the concatenated copy would be dead memory and -- with
                    # more slices -- the projection would run twice per step
                    key = (cat.bits, cat.group_size, cat.input_dims, cat.output_dims)
                    if key not in self._qkv_probe:
                        parts = ops.quantized_matmul_partials(torch.zeros((self.batch, cat.input_dims), dtype=self.dtype, device=self.dev), cat)
                        self._qkv_probe[key] = parts is not None and parts[1] <= 8
                    if not self._qkv_probe[key]:
                        continue
                    self.qkv_cat[li] = cat
        self.use_graph = use_graph
        if use_graph:
            self._capture()

    def norm_check(self) -> None:
        """Raise if a chip-wide slice-sum + norm launch of this runner ran out of polls (synchronises: call off the timed path)."""
        if self.norm_state is not None and ops.norm_status(self.norm_state):
            raise RuntimeError("gbx_add_rmsnorm_slices_xcd: a row's partial statistics did not meet (workgroup placement broke the "
                               "blockIdx % 8 <-> XCD bijection); the results of this runner are invalid -- rerun with GBX_NORM_XCD=0")

    # one decode token, all shapes static, position read from device memory
    def _forward(self):
        m = self.model.model
        pos = self.pos
        B = self.batch
        h = m.embed_tokens(self.tok)[:, None].contiguous()       # [B, 1, D]
        mask = None
        delta = None                                             # residual branch not yet added to h
        for li, (layer, c) in enumerate(zip(m.layers, self.caches)):
            at = layer.self_attn
            norm = layer.input_layernorm
            # residual add of the previous block fused with this norm (gbx_add_rmsnorm); plain norm for the first layer
            if delta is None:
                x = ops.rmsnorm(h, norm.weight, norm.eps)
            elif isinstance(delta, tuple):           # fp32 slices of the previous block's k-split down projection
                x = ops.add_rmsnorm_slices(h, delta[0], delta[1], norm.weight, norm.eps, self.norm_state)
            else:
                x = ops.add_rmsnorm(h, delta, norm.weight, norm.eps)
            qn, kn = getattr(at, "q_norm", None), getattr(at, "k_norm", None)       # Qwen3: per-head norms before RoPE
            cat = self.qkv_cat.get(li)
            parts = ops.quantized_matmul_partials(x.reshape(B, -1), cat) if (cat is not None and c.keys.is_contiguous() and c.values.is_contiguous()) else None
            if parts is not None and parts[1] <= 8:
                # one k-split GEMM over q|k|v, its slices summed inside the attention launch
                o = torch.empty((B, self.n_heads * self.hd), dtype=self.dtype, device=self.dev)
                ops.attn_decode_slices(parts[0], parts[1], c.keys, c.values, at.rope.inv_freq, pos, o, self.n_heads, self.n_kv, self.hd,
                                       getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional,
                                       qn.weight if qn is not None else None, kn.weight if kn is not None else None,
                                       qn.eps if qn is not None else 0.0)
                q = None
            else:
                q, k, v = (t.reshape(B, -1) for t in ops.quantized_matmul_multi(x, (at.q_proj, at.k_proj, at.v_proj)))
            full_rope = at.rope.inv_freq.numel() * 2 == self.hd and c.keys.is_contiguous() and c.values.is_contiguous()
            if q is None:
                pass                                 # (attention already done from the slices)
            elif full_rope and self.hd in (64, 128) and type(at.rope).__name__ in ("RoPE", "Llama3RoPE"):
                # one launch per layer: (q / k norm +) RoPE(q, k) + cache append + softmax(q k^T) v for every sequence
                o = torch.empty_like(q)
                ops.attn_decode_batched(q, k, v, c.keys, c.values, at.rope.inv_freq, pos, o, self.n_heads, self.n_kv, self.hd,
                                        getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional,
                                        qn.weight if qn is not None else None, kn.weight if kn is not None else None,
                                        qn.eps if qn is not None else 0.0)
            else:
                if qn is not None:
                    q = qn(q.reshape(B, self.n_heads, self.hd)).reshape(B, -1)
                    k = kn(k.reshape(B, self.n_kv, self.hd)).reshape(B, -1)
                if full_rope:
                    ops.rope_append(q, k, v, c.keys, c.values, at.rope.inv_freq, pos, self.n_heads, self.n_kv, self.hd,
                                    getattr(at.rope, "scale", 1.0), at.rope.traditional)
                else:                                                # partial-width RoPE: torch glue
                    q = _rope_at(q.reshape(B, 1, self.n_heads, self.hd).transpose(1, 2), at.rope, pos).transpose(1, 2)
                    kr = _rope_at(k.reshape(B, 1, self.n_kv, self.hd).transpose(1, 2), at.rope, pos)
                    c.keys.index_copy_(2, pos, kr)
                    c.values.index_copy_(2, pos, v.reshape(B, 1, self.n_kv, self.hd).transpose(1, 2))
                # grouped-query attention over the whole (static) cache buffer without materialising repeated K/V:
                # q [B, n_kv, rep, hd] x K^T [B, n_kv, hd, cap]; positions past `pos` are masked
                if mask is None:
                    mask = torch.where(self.ar[None, None, None, :] <= pos, 0.0, float("-inf"))
                rep = self.n_heads // self.n_kv
                qg = q.reshape(B, self.n_kv, rep, self.hd)
                sc = torch.matmul(qg, c.keys.transpose(-1, -2)).float() * at.scale + mask
                o = torch.matmul(torch.softmax(sc, dim=-1).to(self.dtype), c.values)     # [B, n_kv, rep, hd]
            norm = layer.post_attention_layernorm
            parts = ops.quantized_matmul_partials(o.reshape(B, -1), at.o_proj) if (B > 1 and self.tp_world == 1) else None
            if parts is not None:                    # k split across workgroups, summed inside the add + norm launch
                x = ops.add_rmsnorm_slices(h, parts[0], parts[1], norm.weight, norm.eps, self.norm_state)
            else:
                o = at.o_proj(o.reshape(B, 1, -1))
                if self.tp_world > 1:
                    all_reduce_(o, at.tp_group)
                x = ops.add_rmsnorm(h, o.reshape(h.shape).contiguous(), norm.weight, norm.eps)
            mlp = layer.mlp
            if B > 1 and hasattr(mlp, "gate_proj"):
                act = ops.quantized_swiglu(x, mlp.gate_proj, mlp.up_proj)              # gate | up + SwiGLU: one launch
                if act is None:
                    g, u = ops.quantized_matmul_multi(x, (mlp.gate_proj, mlp.up_proj))
                    act = ops.swiglu(g, u)
                if getattr(mlp, "tp_world", 1) > 1:  # row-sharded gate / up need no exchange; the K-split down_proj partials meet here
                    delta = mlp.down_proj(act).reshape(h.shape).contiguous()
                    all_reduce_(delta, mlp.tp_group)
                else:
                    # long rows: k split across workgroups; the summing launch is folded into the next add + norm
                    delta = ops.quantized_matmul_partials(act, mlp.down_proj)
                    if delta is None:
                        delta = mlp.down_proj(act).reshape(h.shape)
            else:
                delta = mlp(x).reshape(h.shape).contiguous()
        if isinstance(delta, tuple):
            out = ops.add_rmsnorm_slices(h, delta[0], delta[1], m.norm.weight, m.norm.eps, self.norm_state)
        else:
            h = h + delta
            out = ops.rmsnorm(h, m.norm.weight, m.norm.eps)
        return ops.dense_linear(out, self.model.head_weight())[:, -1, :]   # [B, V]: the head streamed once for the whole batch

    def _capture(self):
        chain = getattr(self, "chain", False)
        safe = min(max(self.caches[0].offset, 0), self.capacity - 1)
        torch.cuda.synchronize(self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):                                   # warm-up outside capture (lazy inits, RCCL communicators)
                self.pos.fill_(safe - 1 if chain else safe)      # a chained step advances the position itself
                self._forward()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = self._forward()

    def start_greedy(self, token: torch.Tensor) -> None:
        """Begin an on-device greedy chain whose first input token is ``token``."""
        self._next = token.reshape(-1).clone()

    def step_greedy(self) -> torch.Tensor:
        """One greedy token: feeds the previous argmax back in.  Generic version (argmax by torch).  With a vocabulary-sharded
        head (tensor parallel) the ranks exchange one (value, index) pair per sequence instead of the logits -- and the return
        value is then THIS RANK'S SHARD of the logits ([B, V / world]); callers that need full logits under TP use ``step()``."""
        if getattr(self.model, "tp_vocab", None) is None:
            logits = self.step(self._next)
            self._next = torch.argmax(logits, -1)
            return logits
        local = self._step_local(self._next)                        # [B, V / world]
        lo = self.model.tp_vocab[0]
        val, idx = torch.max(local.float(), -1)
        pair = torch.stack([val, (idx + lo).float()], -1)           # token ids < 2^24: exact in fp32
        world = self.model.tp_world
        allp = torch.empty((world,) + tuple(pair.shape), dtype=pair.dtype, device=pair.device)
        from .tp import all_gather_into
        all_gather_into(allp, pair, getattr(self.model, "tp_group", None))
        best = torch.argmax(allp[..., 0], 0)                        # first maximum = lowest rank = lowest index on ties
        self._next = allp[best, torch.arange(pair.shape[0], device=pair.device), 1].long()
        return local

    def _check_room(self, off: int) -> None:
        # the attention / append kernels index the cache with the device-resident position and do no bounds check
        if off >= self.capacity:
            raise RuntimeError(f"KV cache is full ({off} of {self.capacity} positions): grow it (KVCache.reserve) and build a new runner")

    def step(self, token: torch.Tensor) -> torch.Tensor:
        """Consumes one token id per sequence (device tensor, shape [B]); returns logits [B, V] (a static buffer; with a
        vocabulary-sharded head: the all-gathered logits)."""
        local = self._step_local(token)
        tv = getattr(self.model, "tp_vocab", None)
        if tv is None or local.shape[-1] == tv[2]:          # (the fused runner gathers inside its launch sequence)
            return local
        from .tp import gather_vocab
        return gather_vocab(local, self.model)

    def _step_local(self, token: torch.Tensor) -> torch.Tensor:
        off = self.caches[0].offset
        self._check_room(off)
        self.tok.copy_(token.reshape(-1))
        self.pos.fill_(off)
        if self.use_graph:
            self.graph.replay()
        else:
            with torch.no_grad():
                self.logits = self._forward()
        for c in self.caches:
            c.offset = off + 1
        return self.logits


class GraphDecodeRunner:
    """Family-agnostic single-token decode: the model's OWN forward is captured once in a HIP graph and replayed.

    What makes an eager step uncapturable is only the KV cache (growing views, a host-side offset feeding RoPE): with the
    caches switched to static mode (``KVCache.static_pos`` / ``static_active``: write at a device-resident position,
    attend over the whole buffer under a position mask, RoPE reads the position from device memory) every family's graph
    -- per-head norms, fused projections, expert routing with the expert-indexed GEMV -- replays unchanged."""

    def __init__(self, model, caches: List[KVCache]):
        self.model, self.caches = model, caches
        self.dev = caches[0].keys.device
        self.capacity = caches[0].capacity
        self.batch = caches[0].keys.shape[0]
        self.tok = torch.zeros((self.batch, 1), dtype=torch.long, device=self.dev)
        self.pos = torch.zeros((1,), dtype=torch.long, device=self.dev)
        for c in caches:
            c.static_pos = self.pos
        safe = min(max(caches[0].offset, 0), self.capacity - 1)
        torch.cuda.synchronize(self.dev)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                                   # warm-up outside capture (lazy inits)
                self.pos.fill_(safe)
                self._forward()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = self._forward()

    def _forward(self):
        for c in self.caches:
            c.static_active = True
        try:
            return self.model(self.tok, cache=self.caches)[:, -1, :]
        finally:
            for c in self.caches:
                c.static_active = False

    def step(self, token: torch.Tensor) -> torch.Tensor:
        off = self.caches[0].offset
        if off >= self.capacity:
            raise RuntimeError(f"KV cache is full ({off} of {self.capacity} positions): grow it and build a new runner")
        self.tok.copy_(token.reshape(self.batch, 1))
        self.pos.fill_(off)
        self.graph.replay()
        for c in self.caches:
            c.offset = off + 1
        return self.logits


def graph_decode_step(model, y: torch.Tensor, prompt_cache: List[KVCache]) -> torch.Tensor:
    """``model.decode_step`` hook of the families without a hand-fused block: HIP-graph replay of the model's own forward."""
    c0 = prompt_cache[0]
    if not isinstance(c0, KVCache) or c0.keys is None or not c0.keys.is_cuda:
        return model(y[None], cache=prompt_cache)[:, -1, :]
    if c0.offset + 1 > c0.capacity:                               # grow every layer before (re)capturing: a full cache doubles (_runner_for)
        more = max(c0.step, min(c0.capacity, 32 * c0.step))
        for c in prompt_cache:
            c.reserve(more, c.keys[:, :, :1], c.values[:, :, :1])
    runner = getattr(model, "_decode_runner", None)
    if getattr(model, "_graph_decode_unavailable", False):
        return model(y[None], cache=prompt_cache)[:, -1, :]
    if runner is None or runner.caches is not prompt_cache or runner.capacity != c0.capacity \
            or c0.keys.data_ptr() != getattr(runner, "_kptr", None):
        try:
            runner = GraphDecodeRunner(model, prompt_cache)
        except ops.NotCapturable as e:
            # a forward that cannot be captured (more (token, expert) pairs than the expert-indexed launch covers needs the
            # host-grouped gather): this model decodes eagerly from here on.  Anything else -- out of memory, a HIP error, a bug
            # in the captured forward -- propagates.
            import warnings
            warnings.warn(f"gbx_lm_amd: graph-replayed decode is unavailable for this model ({e}); decoding eagerly")
            for c in prompt_cache:
                c.static_active, c.static_pos = False, None
            model._graph_decode_unavailable = True
            torch.cuda.synchronize()
            return model(y[None], cache=prompt_cache)[:, -1, :]
        runner._kptr = c0.keys.data_ptr()
        model._decode_runner = runner
    return runner.step(y)


def fused_eligible(model) -> bool:
    """The fused HIP decode block needs: no biases, q/k/v (and gate/up) packed alike, head_dim 64/128, full-width RoPE,
    16-bit activations and every (per-rank) K a multiple of 128.  Tensor-parallel shards qualify too."""
    dt = model.model.embed_tokens.weight.dtype
    if dt not in (torch.bfloat16, torch.float16):
        return False
    for blk in model.model.layers:
        at, mlp = blk.self_attn, blk.mlp
        mods = (at.q_proj, at.k_proj, at.v_proj, at.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj)
        if any(m.input_dims % 128 or m.scales is None or m.scales.dtype != dt for m in mods):
            return False
        # biases: none, or q / k / v all three (Qwen2) -- those ride in the attention launch
        qkv_b = [m.bias for m in (at.q_proj, at.k_proj, at.v_proj)]
        if any(m.bias is not None for m in mods[3:]) or (any(b is not None for b in qkv_b) and
                                                         not all(b is not None and b.dtype == dt for b in qkv_b)):
            return False
        sig = lambda m: (m.bits, m.group_size)
        if not (sig(at.q_proj) == sig(at.k_proj) == sig(at.v_proj)) or sig(mlp.gate_proj) != sig(mlp.up_proj):
            return False
        if at.head_dim not in (64, 128) or at.rope.inv_freq.numel() * 2 != at.head_dim:
            return False
        if type(at.rope).__name__ not in ("RoPE", "Llama3RoPE"):          # amplitude-rescaling variants (su / YaRN): generic path
            return False
        qn, kn = getattr(at, "q_norm", None), getattr(at, "k_norm", None)    # Qwen3: per-head norms inside the attention launch
        if (qn is None) != (kn is None) or (qn is not None and (qn.weight.dtype != dt or kn.weight.dtype != dt)):
            return False
        vpl = lambda m: 64 if m.bits == 2 else 32
        if at.q_proj.input_dims // vpl(at.q_proj) > 128 or mlp.gate_proj.input_dims // vpl(mlp.gate_proj) > 128:
            return False
        if any(m.input_dims // vpl(m) > 512 for m in mods):
            return False
    hw = model.head_weight()
    return hw.shape[1] % 8 == 0 and hw.shape[1] <= 8192 and hw.dtype == dt


class FusedDecodeRunner(DecodeRunner):
    """Same contract as DecodeRunner, but the block is 5 HIP launches per layer through the C-ABI fused entry points:
    [RMSNorm + q/k/v GEMV] -> [RoPE + cache append + attention] -> [o_proj GEMV + residual] ->
    [RMSNorm + gate/up GEMV + SwiGLU] -> [down_proj GEMV + residual], then [final RMSNorm + lm_head GEMV]."""

    def __init__(self, model, caches: List[KVCache], use_graph: bool = True):
        dev = next(model.parameters()).device
        dt = model.model.embed_tokens.weight.dtype
        a0 = model.model.layers[0].self_attn
        D = model.args.hidden_size
        I = model.model.layers[0].mlp.gate_proj.output_dims
        self._n_layers = len(model.model.layers)
        self._mostly_4bit = 2 * sum(b.mlp.gate_proj.bits == 4 for b in model.model.layers) >= self._n_layers
        self._mostly_3bit = 2 * sum(b.mlp.gate_proj.bits == 3 for b in model.model.layers) >= self._n_layers or env.get("GBX_DMLP_MX3") == "2"
        mk = lambda n: torch.zeros((n,), dtype=dt, device=dev)
        self.pos = torch.zeros((1,), dtype=torch.long, device=dev)
        nq, nkv = a0.n_heads * a0.head_dim, a0.n_kv_heads * a0.head_dim
        self.qkvb = mk(nq + 2 * nkv)             # q | k | v of the token: one buffer (one-segment q|k|v launch below), three views
        self.hb, self.qb, self.kb, self.vb = mk(D), self.qkvb[:nq], self.qkvb[nq:nq + nkv], self.qkvb[nq + nkv:]
        self.ab, self.actb = mk(a0.n_heads * a0.head_dim), mk(I)
        # TP: the residual stream ping-pongs between hb and pb.  Rank 0 folds the residual into its partial sum
        # (y = stream + W_0 x), the other ranks write bare partials, and the all-reduce of that buffer IS the new stream --
        # no separate residual-add launch between the collective and the next GEMV.
        self.pb = mk(D)
        tp = getattr(model, "tp_world", 1) > 1
        r0 = getattr(model, "tp_rank", 0) == 0
        self.logits_buf = torch.zeros((1, model.head_weight().shape[0]), dtype=dt, device=dev)
        self.keys = torch.zeros((64,), dtype=torch.int64, device=dev)      # sharded (value, index) argmax keys (uint64 bits)
        # generate_step's default-sampler loop as one replay per token (step_gen): logprobs + argmax inside the graph, the token fed back
        # on the device, (token, hand-off status words) packed for ONE copy to the host
        from . import _lib as _l0
        self.lp_buf = torch.zeros_like(self.logits_buf)
        self.lp_ws = torch.zeros((_l0.lib().gbx_logprobs_workspace_bytes(),), dtype=torch.uint8, device=dev)
        self.gen_words = torch.zeros((4,), dtype=torch.int64, device=dev)
        self._gen_next = None               # (cache offset, token tensor) the device-side feedback expects next
        self._tok_view = None
        # vocabulary-sharded head (tensor parallel): every rank's keys, gathered; the full logits for callers of step()
        self.tp_vocab = getattr(model, "tp_vocab", None)
        if self.tp_vocab is not None:
            self.all_keys = torch.zeros((64 * model.tp_world,), dtype=torch.int64, device=dev)
            self.logits_full = torch.zeros((1, self.tp_vocab[2]), dtype=dt, device=dev)
        from . import _lib as _l
        need = _l.lib().gbx_attn_workspace_bytes(1, a0.n_heads, a0.n_kv_heads, a0.head_dim, caches[0].capacity)
        self.attn_ws = torch.empty((need,), dtype=torch.uint8, device=dev) if need else None   # context split of long caches
        self.chain = False                  # True while capturing / replaying the on-device greedy chain
        self.graph_chain = None
        # Short caches (no context split): attention + o_proj as ONE launch whose partial sums land in a fixed-point
        # accumulator; the gate|up launch forms h + o in its prologue and the down launch clears the accumulator -- four
        # launches per block instead of five (gbx_attn_oproj_decode).  Opt-in (GBX_FUSE_AO=1): measured neutral -- the in-launch
        # hand-off costs what the removed boundary cost (profiles/r03_chain_floor.txt: 750-753 vs 750-762 tok/s).
        import os
        self.acc = torch.zeros((D + a0.n_kv_heads * ops.ATTN_OPROJ_HANDOFF_WORDS,), dtype=torch.int64, device=dev)   # o_proj sums | hand-off words
        self.fuse_ao = (env.flag("GBX_FUSE_AO") and not tp and self.attn_ws is None and
                        all(getattr(b.self_attn, "q_norm", None) is None and b.self_attn.q_proj.bias is None and
                            ops.attn_oproj_supported(b.self_attn.n_heads, b.self_attn.n_kv_heads, b.self_attn.head_dim, b.self_attn.o_proj) and
                            b.mlp.gate_proj.group_size >= 64 and
                            b.mlp.gate_proj.input_dims // (64 if b.mlp.gate_proj.bits == 2 else 32) <= 128
                            for b in model.model.layers))
        # Round 5: RMSNorm + q|k|v + RoPE / cache append + attention + o_proj as ONE launch per block (gbx_dattn_forward: one kv head per
        # XCD, hand-offs inside the XCD's L2, o_proj K-split by kv group into a fixed-point accumulator); the gate|up launch forms
        # h + o in its prologue, the down launch clears the accumulator -- three launches per block instead of five.
        # Long caches (the context split's territory): the launch's cooperative sweep covers them when a kv group has <= 4 query heads.
        # (long caches with 1 or 2 query heads per kv head have no cooperative sweep: ONE workgroup per head would pull the whole context)
        # A model whose hand-offs timed out once (decode_fallback) stays on the five-launch block; so does a device that fails the
        # layout check (partition mode, CU mask, a dispatcher that does not deal workgroups round-robin over the XCDs).
        self.use_dattn = (env.flag("GBX_DATTN") and not tp and not self.fuse_ao and env.flag("GBX_DGEMV") and env.flag("GBX_QKV_CAT") and
                          (self.attn_ws is None or a0.n_heads // a0.n_kv_heads in (3, 4)) and
                          not getattr(model, "_fused_handoff_disabled", False) and ops.xcd_layout_ok(dev))
        self.use_dmlp = self.use_dattn and env.flag("GBX_DMLP")
        # Round 6: the whole 2-bit block as ONE launch (gbx_dblock_forward) while the context is short enough for the head's own workgroup
        # to sweep it; beyond gbx_dblock_max_context() cached positions the same descriptors replay as the two launches above (a second
        # captured graph, chosen on the host by the position it knows anyway).
        self.use_dblock = self.use_dmlp and env.flag("GBX_DBLOCK")
        self.db = []                        # per layer: the one-launch descriptor, or None
        self.db_state = None
        self.db_max_ctx = 0
        self.long_mode = False              # True while capturing / running the graph of contexts beyond db_max_ctx
        self._graphs = {}                   # (chain, long) -> (graph, logits)
        self.da = []                        # per layer: (dattn, gate|up, down) or (dattn, dmlp) descriptors, or None (the five-launch block)
        self.da_pre = {}                    # layer -> (x, sums): pending fixed-point sums to finish into hb before a five-launch block
        self.da_post = None                 # the same after the last block
        self.acc_o = self.acc_d = self.hn = self.da_state = self.dm_state = None
        self._pending = None
        self.descs = []
        self.descs_ao = []
        self.dg = []                        # per layer: [qkv, o, gate|up, down] dgemv descriptors (None: the launch stays on gbx_qlinear_fused_forward)
        self.dg_keep = []
        self.use_dg = env.flag("GBX_DGEMV")
        self.use_ssq = env.flag("GBX_DG_SSQ") and not tp
        self.ssq_prev = None                # partial sums of h^2 left by the previous layer's down_proj launch
        self.packed = []                    # load-time (scale | zero) companions, kept alive for the descriptors
        for blk in model.model.layers:
            at, mlp = blk.self_attn, blk.mlp
            eps = blk.input_layernorm.eps
            even = all(m.output_dims % 2 == 0 for m in (at.q_proj, at.k_proj, at.v_proj, at.o_proj, mlp.gate_proj, mlp.down_proj))
            pk = lambda *mods: torch.cat([ops.pack_scales_zeros(m) for m in mods]) if even else None
            p_qkv, p_o, p_dn = pk(at.q_proj, at.k_proj, at.v_proj), pk(at.o_proj), pk(mlp.down_proj)
            # q|k|v as ONE matrix (rows concatenated once at load time, +7 MB per 2-bit 8B layer): a one-segment launch skips the
            # per-row segment location of the three-segment form (the launch is bound by per-batch instruction overhead)
            qkv_mods, qkv_ys = (at.q_proj, at.k_proj, at.v_proj), (self.qb, self.kb, self.vb)
            if (env.flag("GBX_QKV_CAT") and p_qkv is not None and nq % 16 == 0 and nkv % 16 == 0 and
                    len({(m.bits, m.group_size) for m in qkv_mods}) == 1 and all(m.bias is None for m in qkv_mods)):
                cat = ops.qkv_concat(at.q_proj, at.k_proj, at.v_proj)
                self.packed.append((cat,))
                qkv_mods, qkv_ys = (cat,), (self.qkvb,)
            p_gu = ops.pack_scales_zeros(mlp.gate_proj, mlp.up_proj) if even else None
            self.packed.append((p_qkv, p_o, p_gu, p_dn))
            self.descs.append((
                ops.make_fused_desc(self.hb, qkv_mods, qkv_ys,
                                    norm_weight=blk.input_layernorm.weight, eps=eps, packed=p_qkv, bias_elsewhere=True),
                ops.make_fused_desc(self.ab, (at.o_proj,), (self.pb,), residual=self.hb if r0 else None, packed=p_o) if tp else
                ops.make_fused_desc(self.ab, (at.o_proj,), (self.hb,), residual=self.hb, packed=p_o),
                ops.make_fused_desc(self.pb if tp else self.hb, (mlp.gate_proj, mlp.up_proj), (self.actb,),
                                    norm_weight=blk.post_attention_layernorm.weight, eps=blk.post_attention_layernorm.eps,
                                    swiglu=True, packed=p_gu),
                ops.make_fused_desc(self.actb, (mlp.down_proj,), (self.hb,), residual=self.pb if r0 else None, packed=p_dn) if tp else
                ops.make_fused_desc(self.actb, (mlp.down_proj,), (self.hb,), residual=self.hb, packed=p_dn),
            ))
            # Round 4: the four launches on the decode-block GEMV (gbx_dgemv_forward: compact kernel arguments, 2 / 4 rows per
            # batch, vector epilogue) wherever it covers the shape; the RMSNorm statistic of h is carried from the launch that
            # writes h (o_proj / down_proj epilogue, one partial per workgroup) to the launch that normalises it (GBX_DG_SSQ).
            ssq_prev_layer = self.ssq_prev
            self.dg.append(self._dgemv_descs(blk, qkv_mods[0] if len(qkv_mods) == 1 else None, qkv_ys[0], tp, r0, len(self.dg)))
            self.da.append(self._fused_descs(len(self.da), blk, qkv_mods[0] if len(qkv_mods) == 1 else None, caches[len(self.da)], ssq_prev_layer))
            self.db.append(self._block_desc(blk, qkv_mods[0] if len(qkv_mods) == 1 else None, self.da[-1]))
            if self.fuse_ao and p_gu is not None:
                self.descs_ao.append((
                    # x = hb + round(acc) formed in the prologue, the new stream written to pb once
                    ops.make_fused_desc(self.hb, (mlp.gate_proj, mlp.up_proj), (self.actb,),
                                        norm_weight=blk.post_attention_layernorm.weight, eps=blk.post_attention_layernorm.eps,
                                        swiglu=True, packed=p_gu, x_acc=self.acc[:D], x_out=self.pb),
                    # hb = pb + down(act); clears the accumulator for the next block
                    ops.make_fused_desc(self.actb, (mlp.down_proj,), (self.hb,), residual=self.pb, packed=p_dn, zero_buf=self.acc),
                ))
            else:
                self.fuse_ao = False
        self.da_post, self._pending = self._pending, None
        super().__init__(model, caches, use_graph=False)       # (graphs are captured per mode, the first time a step needs one)
        self.use_graph = use_graph           # (each mode's graph is captured by the first step that needs it, with its long-context twin)

    def _dgemv_descs(self, blk, qkv_cat, qkv_y, tp: bool, r0: bool, li: int):
        """dgemv descriptors of one block's four launches (None where the shape is not covered or dgemv is switched off)."""
        from . import _lib as _l
        none = [None, None, None, None]
        if not self.use_dg or self.fuse_ao:
            return none
        at, mlp = blk.self_attn, blk.mlp
        dev = self.hb.device
        out = list(none)

        def plan(m, epi, norm=False):
            return ops.dgemv_plan(m.output_dims, m.input_dims, m.bits, m.group_size, epi, norm)

        def ssq_buf():
            t = torch.zeros((_l.GBX_DG_SSQ_SLOTS,), dtype=torch.float32, device=dev)
            self.dg_keep.append(t)
            return t

        ssq_o = ssq_dn = None
        # o_proj (+ residual)
        res_o = (self.hb if r0 else None) if tp else self.hb
        pl = plan(at.o_proj, _l.GBX_DG_RESIDUAL if res_o is not None else _l.GBX_DG_PLAIN) if at.o_proj.bias is None else None
        if pl is not None:
            pk = ops.dgemv_pack(at.o_proj, pl[0])
            if self.use_ssq and pl[1] > 0:
                ssq_o = ssq_buf()
            out[1] = ops.make_dgemv_desc(self.ab, at.o_proj, self.pb if tp else self.hb, pk, pl[0], residual=res_o, ssq_out=ssq_o)
            self.dg_keep.append(pk)
        # down_proj (+ residual)
        res_d = (self.pb if r0 else None) if tp else self.hb
        pl = plan(mlp.down_proj, _l.GBX_DG_RESIDUAL if res_d is not None else _l.GBX_DG_PLAIN) if mlp.down_proj.bias is None else None
        if pl is not None:
            pk = ops.dgemv_pack(mlp.down_proj, pl[0])
            if self.use_ssq and pl[1] > 0:
                ssq_dn = ssq_buf()
            out[3] = ops.make_dgemv_desc(self.actb, mlp.down_proj, self.hb, pk, pl[0], residual=res_d, ssq_out=ssq_dn)
            self.dg_keep.append(pk)
        # q|k|v as one matrix: RMSNorm prologue; its statistic comes from the previous block's down_proj launch
        if qkv_cat is not None:
            pl = plan(qkv_cat, _l.GBX_DG_PLAIN, True)
            if pl is not None:
                pk = ops.dgemv_pack(qkv_cat, pl[0])
                out[0] = ops.make_dgemv_desc(self.hb, qkv_cat, qkv_y, pk, pl[0], norm_weight=blk.input_layernorm.weight,
                                             eps=blk.input_layernorm.eps, ssq_in=self.ssq_prev)
                self.dg_keep.append(pk)
        # gate | up + SwiGLU: RMSNorm prologue; statistic from this block's o_proj launch
        if mlp.gate_proj.bias is None and mlp.up_proj.bias is None:
            pl = plan(mlp.gate_proj, _l.GBX_DG_SWIGLU, True)
            if pl is not None:
                pk = ops.dgemv_pack(mlp.gate_proj, pl[0], mlp.up_proj)
                out[2] = ops.make_dgemv_desc(self.pb if tp else self.hb, mlp.gate_proj, self.actb, pk, pl[0], mod_up=mlp.up_proj,
                                             norm_weight=blk.post_attention_layernorm.weight, eps=blk.post_attention_layernorm.eps,
                                             ssq_in=ssq_o if out[1] is not None else None)
                self.dg_keep.append(pk)
        self.ssq_prev = ssq_dn if out[3] is not None else None
        return out

    def _fused_descs(self, li, blk, qkv_cat, cache, ssq_prev):
        """Descriptors of one block on the fused launches, or None (the five-launch block):
        (fused attention, gate|up with the accumulator prologue, down_proj)  -- three launches, or
        (fused attention, fused MLP)                                        -- two launches (gbx_dmlp_forward: 2-bit Llama-3-8B MLP shapes).
        The stream enters a block either as a vector (``self._pending is None``) or as (vector, fixed-point down_proj sums) left
        by the previous block's fused MLP launch; the fused attention launch takes both forms.  Accumulators alternate by block
        parity so that a launch never clears a buffer another launch of the same kernel still reads or adds to."""
        from . import _lib as _l
        at, mlp = blk.self_attn, blk.mlp
        pending, self._pending = self._pending, None
        D = self.hb.numel()
        dev = self.hb.device
        ok = (self.use_dattn and qkv_cat is not None and getattr(at, "q_norm", None) is None and at.o_proj.bias is None and
              mlp.gate_proj.bias is None and mlp.up_proj.bias is None and mlp.down_proj.bias is None and
              cache.keys is not None and cache.keys.shape[0] == 1 and cache.keys.is_contiguous() and cache.values.is_contiguous() and
              ops.dattn_supported(D, at.n_heads, at.n_kv_heads, at.head_dim, qkv_cat, at.o_proj))
        pl_gu = pl_dn = None
        if ok:
            pl_gu = ops.dgemv_plan(mlp.gate_proj.output_dims, mlp.gate_proj.input_dims, mlp.gate_proj.bits, mlp.gate_proj.group_size,
                                   _l.GBX_DG_SWIGLU, True)
            pl_dn = ops.dgemv_plan(mlp.down_proj.output_dims, mlp.down_proj.input_dims, mlp.down_proj.bits, mlp.down_proj.group_size,
                                   _l.GBX_DG_RESIDUAL, False)
            ok = pl_gu is not None and pl_dn is not None and self.dg[-1][3] is not None
        if not ok:
            if pending is not None:          # the five-launch block takes a plain vector: finish the pending sums into hb first
                self.da_pre[li] = (pending[0], pending[1])
            return None
        if self.acc_o is None:
            # o_proj / down_proj sums by block parity: [D + 1] views of two rows of D + 2 words (16-byte aligned rows)
            self.acc_o_flat = torch.zeros((2 * (D + 2),), dtype=torch.int64, device=dev)
            self.acc_d_flat = torch.zeros((2 * (D + 2),), dtype=torch.int64, device=dev)
            self.acc_o = [self.acc_o_flat[q * (D + 2): q * (D + 2) + D + 1] for q in range(2)]
            self.acc_d = [self.acc_d_flat[q * (D + 2): q * (D + 2) + D + 1] for q in range(2)]
            self.hn = torch.zeros((D,), dtype=self.hb.dtype, device=dev)             # the stream a fused attention launch forms from (x, sums)
            self.da_state = ops.dattn_state(dev)
            self.dm_state = ops.dattn_state(dev)
        p = li & 1
        pk_qkv = ops.dgemv_pack(qkv_cat, 2)
        self.dg_keep.append(pk_qkv)
        x_in, acc_in = (pending if pending is not None else (self.hb, None))
        d_a = ops.make_dattn_desc(x_in, blk.input_layernorm.weight, blk.input_layernorm.eps, qkv_cat, pk_qkv, cache.keys, cache.values,
                                  at.rope.inv_freq, self.pos_buf(), at.o_proj, self.acc_o[p], self.da_state, at.n_heads, cache.capacity,
                                  getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional,
                                  ssq_in=ssq_prev if acc_in is None else None, x_acc=acc_in, x_out=self.hn if acc_in is not None else None,
                                  zero_buf=self.acc_d[p])
        h_buf = self.hn if acc_in is not None else x_in          # the stream before o_proj's output is added
        # (the fused MLP launch exists for 3 / 4 bits too -- dmlpv.hip -- but measured slower than the two launches it replaces: opt-in)
        mx4 = (mlp.gate_proj.bits == 4 and mlp.down_proj.bits == 4 and mlp.gate_proj.group_size == 128 and mlp.down_proj.group_size == 128 and
               D in (3072, 4096) and env.flag("GBX_DMLP_MX4") and
               # (round 6: the matrix-core launch takes 4-bit rows as pairs of 2-bit fields.  Not for the LAST block: its down launch
               #  writes the stream the head reads, where the fused launch would need one more launch to finish its sums)
               li != self._n_layers - 1 and self._mostly_4bit)
        # (an isolated 4-bit block between three-launch blocks gains nothing: the block behind it then starts from (vector, sums) instead of a
        #  finished stream with its statistic carried -- measured level on the S22 mix, +1.9 % on the all-4-bit strategy)
        mx3 = (mlp.gate_proj.bits == 3 and mlp.down_proj.bits == 3 and mlp.gate_proj.group_size == 128 and mlp.down_proj.group_size == 128 and
               D in (3072, 4096) and env.flag("GBX_DMLP_MX3") and li != self._n_layers - 1 and self._mostly_3bit)
        # (3-bit rows repacked once as a 2-bit and a 1-bit plane: +2.9 % on an all-3-bit 8B model, +3.1 % at the 3.2-3B geometry; on the S22 mix
        #  -- three 3-bit blocks between one-launch blocks -- the token got 1.5 % SLOWER although the three launches themselves gained
        #  11 us: every other launch of the token ran ~1.5 % longer in that configuration (profiles/r06_dmlp_mx3.txt); hence the same rule as for 4 bits)
        use_mlp = (self.use_dmlp and (mlp.gate_proj.bits == 2 or mx4 or mx3 or env.flag("GBX_DMLP_VEC")) and
                   ops.dmlp_supported(D, mlp.gate_proj, mlp.up_proj, mlp.down_proj, self.hb.dtype) and
                   blk.post_attention_layernorm.weight.dtype == self.hb.dtype)
        if use_mlp:
            pk_gu = ops.dgemv_pack(mlp.gate_proj, 2, mlp.up_proj)
            pk_dn = ops.dgemv_pack(mlp.down_proj, 2)
            self.dg_keep += [pk_gu, pk_dn]
            d_m = ops.make_dmlp_desc(h_buf, self.acc_o[p], self.pb, blk.post_attention_layernorm.weight, blk.post_attention_layernorm.eps,
                                     mlp.gate_proj, mlp.up_proj, mlp.down_proj, pk_gu, pk_dn, self.acc_d[p], self.dm_state,
                                     zero_buf=self.acc_o[1 - p])
            self._pending = (self.pb, self.acc_d[p])
            self.ssq_prev = None                                 # no down launch: no carried statistic for the next block
            return d_a, d_m
        pk_gu = ops.dgemv_pack(mlp.gate_proj, pl_gu[0], mlp.up_proj)
        pk_dn = ops.dgemv_pack(mlp.down_proj, pl_dn[0])
        self.dg_keep += [pk_gu, pk_dn]
        d_gu = ops.make_dgemv_desc(h_buf, mlp.gate_proj, self.actb, pk_gu, pl_gu[0], mod_up=mlp.up_proj,
                                   norm_weight=blk.post_attention_layernorm.weight, eps=blk.post_attention_layernorm.eps,
                                   x_acc=self.acc_o[p], x_out=self.pb)
        # the statistic of the stream this launch writes goes to the next block's norm'd launch exactly as in the five-launch block
        d_dn = ops.make_dgemv_desc(self.actb, mlp.down_proj, self.hb, pk_dn, pl_dn[0], residual=self.pb,
                                   ssq_out=self.ssq_prev, zero_buf=self.acc_o_flat)
        return d_a, d_gu, d_dn

    def _block_desc(self, blk, qkv_cat, fused):
        """The one-launch descriptor of a block whose (fused attention, fused MLP) pair exists and whose shapes gbx_dblock covers."""
        if not self.use_dblock or fused is None or len(fused) != 2 or qkv_cat is None:
            return None
        at, mlp = blk.self_attn, blk.mlp
        if not ops.dblock_supported(self.hb.numel(), at.n_heads, at.n_kv_heads, at.head_dim, qkv_cat, at.o_proj, mlp.gate_proj, mlp.up_proj,
                                    mlp.down_proj, self.hb.dtype):
            return None
        if self.db_state is None:
            self.db_state = ops.dblock_state(self.hb.device)
            self.db_max_ctx = ops.dblock_max_context()
        return ops.make_dblock_desc(fused[0], fused[1], self.db_state)

    def pos_buf(self) -> torch.Tensor:
        return self.pos

    # ---- captured graphs by (greedy chain?, long context?): the one-launch block up to db_max_ctx cached positions, the two-launch block
    #      (cooperative sweep) beyond; each is captured the first time a step needs it, at the position the step is at ----
    def _wants_long(self, off: int) -> bool:
        return self.db_state is not None and off > self.db_max_ctx

    def _graph_for(self, chain: bool, long: bool):
        """(graph, logits) of the mode, capturing it first if need be.  The warm-up passes of a capture run real steps at the current
        position: they write cache rows >= the current offset only and leave every accumulator zero; keys / position / offsets are restored."""
        slot = (chain, long)
        if slot not in self._graphs:
            off = self.caches[0].offset
            saved = (self.graph, self.logits, self.chain, self.long_mode)
            keys0, pos0, tok0 = self.keys.clone(), self.pos.clone(), self.tok.clone()
            akeys0 = self.all_keys.clone() if self.tp_vocab is not None else None
            self.chain, self.long_mode = chain, long
            self._capture()
            self._graphs[slot] = (self.graph, self.logits)
            self.graph, self.logits, self.chain, self.long_mode = saved
            if self.graph is None:                                   # (what callers and tests look at: the first captured graph and the static logits)
                self.graph, self.logits = self._graphs[slot]
            for c in self.caches:
                c.offset = off
            self.keys.copy_(keys0)
            self.pos.copy_(pos0)
            self.tok.copy_(tok0)
            if akeys0 is not None:
                self.all_keys.copy_(akeys0)
        return self._graphs[slot]

    def _capture_other_mode(self, chain: bool) -> None:
        """A cache that can outgrow the one-launch block's context limit will need the two-launch graph too: capture it NOW rather than in the
        middle of a generation (a capture costs ~10 ms: two warm-up tokens + the instantiate)."""
        if self.db_state is not None and self.capacity > self.db_max_ctx and self.caches[0].offset <= self.db_max_ctx:
            self._graph_for(chain, True)

    def step_gen(self, token: torch.Tensor):
        """One step of generate_step's default-sampler loop (utils.py:285,305-306,323-338) as ONE graph replay: the block launches, the
        head, logprobs = logits - logsumexp(logits), token = argmax(logprobs) -- and the token fed back on the device, so consecutive calls
        that pass the token tensor this method returned need no host-side copy or position update at all.  Returns (token int64 [1],
        logprobs [1, V], packed int64 [4] = (token, hand-off status words)): views of STATIC buffers, overwritten by the next call --
        the caller copies what it keeps."""
        off = self.caches[0].offset
        self._check_room(off)
        long = self._wants_long(off)
        if self._tok_view is None:
            self._tok_view = self.tok.view(-1)
        fed_back = self._gen_next is not None and self._gen_next[0] == off and token is self._gen_next[1]
        if self.use_graph:
            first = ("gen", long) not in self._graphs
            graph, _ = self._graph_for("gen", long)
            if first:
                self._capture_other_mode("gen")
            if not fed_back:
                self.tok.copy_(token.reshape(self.tok.shape))
                self.pos.fill_(off - 1)                               # the embedding launch advances it to `off`
            graph.replay()
        else:
            if not fed_back:
                self.tok.copy_(token.reshape(self.tok.shape))
                self.pos.fill_(off - 1)
            self.chain, self.long_mode = "gen", long
            with torch.no_grad():
                self._forward()
            self.chain, self.long_mode = False, False
        for c in self.caches:
            c.offset = off + 1
        self._gen_next = (off + 1, self._tok_view)
        return self._tok_view, self.lp_buf, self.gen_words

    def _step_local(self, token: torch.Tensor) -> torch.Tensor:
        off = self.caches[0].offset
        self._check_room(off)
        self._gen_next = None
        long = self._wants_long(off)
        if self.use_graph:
            first = (False, long) not in self._graphs
            graph, out = self._graph_for(False, long)
            if first:
                self._capture_other_mode(False)
            self.tok.copy_(token.reshape(-1))
            self.pos.fill_(off)
            graph.replay()
        else:
            self.tok.copy_(token.reshape(-1))
            self.pos.fill_(off)
            self.long_mode = long
            with torch.no_grad():
                out = self.logits = self._forward()
            self.long_mode = False
        for c in self.caches:
            c.offset = off + 1
        return out

    def fault_words(self):
        """The status words of this runner's fused launches as one-element int32 device views (non-zero: a hand-off timed out and the
        stream was poisoned).  generate_step copies them to the host next to every token and calls decode_fallback on a non-zero one."""
        words = [ops.dattn_status_word(st) for st in (self.da_state, self.dm_state) if st is not None]
        if self.db_state is not None:
            words.append(ops.dblock_status_word(self.db_state))
        return words

    def dattn_check(self) -> None:
        """Raise if a fused attention launch of this runner timed out on a hand-off (synchronises: call off the timed path)."""
        for st in (self.da_state, self.dm_state, self.db_state):
            code = (ops.dblock_status(st) if st is self.db_state else ops.dattn_status(st)) if st is not None else 0
            if code:
                raise RuntimeError(f"gbx_dattn_forward: hand-off {code} timed out (workgroup placement broke the one-kv-group-per-XCD "
                                   "bijection, or a member never ran); the results of this runner are invalid -- rerun with GBX_DATTN=0")

    def _forward(self):
        m = self.model.model
        dev = self.dev
        # token -> residual stream on the device (also clears the argmax keys and, in a greedy chain, takes the token
        # from them and advances the position: no host involvement between tokens)
        if self.chain and self.tp_vocab is not None:
            ops.embed_next_token_sharded(self.all_keys, self.keys, self.tok, self.pos, m.embed_tokens.weight, self.hb, advance_pos=True)
        else:
            ops.embed_next_token(self.keys, self.tok, self.pos, m.embed_tokens.weight, self.hb, from_keys=self.chain is True,
                                 advance_pos=bool(self.chain))
        for li, (blk, c, (d_qkv, d_o, d_gu, d_dn)) in enumerate(zip(m.layers, self.caches, self.descs)):
            at = blk.self_attn
            g_qkv, g_o, g_gu, g_dn = self.dg[li]
            run = lambda g, d: ops.dgemv(g, dev) if g is not None else ops.qlinear_fused(d, dev)
            if li in self.da_pre:            # pending fixed-point sums -> hb (every accumulator is zero afterwards)
                xs, accs = self.da_pre[li]
                ops.acc_residual(xs, accs, self.hb, zero=[t for t in (self.acc_d[0], self.acc_d[1], self.acc_o[0], self.acc_o[1]) if t.data_ptr() != accs.data_ptr()])
            if self.db[li] is not None and not self.long_mode:
                ops.dblock(self.db[li], dev)
                continue
            if self.da[li] is not None:
                ops.dattn(self.da[li][0], dev)
                if len(self.da[li]) == 2:
                    ops.dmlp(self.da[li][1], dev)
                else:
                    ops.dgemv(self.da[li][1], dev)
                    ops.dgemv(self.da[li][2], dev)
                continue
            run(g_qkv, d_qkv)
            if self.fuse_ao:
                d_gu2, d_dn2 = self.descs_ao[li]
                ops.attn_oproj_decode(self.qb, self.kb, self.vb, c.keys, c.values, at.rope.inv_freq, self.pos, at.o_proj, self.acc,
                                      at.n_heads, at.n_kv_heads, at.head_dim, c.capacity, getattr(at.rope, "scale", 1.0), at.scale,
                                      at.rope.traditional)
                ops.qlinear_fused(d_gu2, dev)
                ops.qlinear_fused(d_dn2, dev)
                continue
            qn = getattr(at, "q_norm", None)
            if qn is not None or at.q_proj.bias is not None or self.attn_ws is not None:
                # Qwen3 (per-head q / k norms) and Qwen2 (q / k / v biases): the grouped kernel takes both in its prologue;
                # long caches: the same kernel with the context split over workgroups (scratch from the runner)
                ops.attn_decode_batched(self.qb.view(1, -1), self.kb.view(1, -1), self.vb.view(1, -1), c.keys, c.values,
                                        at.rope.inv_freq, self.pos, self.ab.view(1, -1), at.n_heads, at.n_kv_heads, at.head_dim,
                                        getattr(at.rope, "scale", 1.0), at.scale, at.rope.traditional,
                                        qn.weight if qn is not None else None, at.k_norm.weight if qn is not None else None,
                                        qn.eps if qn is not None else 0.0, at.q_proj.bias, at.k_proj.bias, at.v_proj.bias,
                                        workspace=self.attn_ws)
            else:
                ops.attn_decode(self.qb, self.kb, self.vb, c.keys, c.values, at.rope.inv_freq, self.pos, self.ab,
                                at.n_heads, at.n_kv_heads, at.head_dim, c.capacity, getattr(at.rope, "scale", 1.0), at.scale,
                                at.rope.traditional)
            run(g_o, d_o)
            if self.tp_world > 1:            # K-split layer: partial sums (+ rank 0's residual) meet over RCCL in pb
                all_reduce_(self.pb, at.tp_group)
            run(g_gu, d_gu)
            run(g_dn, d_dn)
            if self.tp_world > 1:            # ... and the stream returns to hb
                all_reduce_(self.hb, blk.mlp.tp_group)
        if self.da_post is not None:         # the last block's down_proj sums -> hb
            xs, accs = self.da_post
            ops.acc_residual(xs, accs, self.hb, zero=[t for t in (self.acc_d[0], self.acc_d[1], self.acc_o[0], self.acc_o[1]) if t.data_ptr() != accs.data_ptr()])
        ops.dense_gemv(self.hb, self.model.head_weight(), out=self.logits_buf.view(-1), norm_weight=m.norm.weight,
                       eps=m.norm.eps, argmax_keys=self.keys, index_offset=self.tp_vocab[0] if self.tp_vocab is not None else 0)
        if self.tp_vocab is None:
            if self.chain == "gen":          # utils.py:305-306,285 inside the graph; the token lands where the next replay's embedding reads it
                ops.logprobs_argmax(self.logits_buf, out=self.lp_buf, token=self.tok.view(-1), workspace=self.lp_ws,
                                    status=self.fault_words(), packed=self.gen_words)
            return self.logits_buf
        from .tp import all_gather_into
        grp = getattr(self.model, "tp_group", None)
        if self.chain:                       # greedy chain: 64 keys per rank are all the next step needs
            all_gather_into(self.all_keys, self.keys, grp)
            return self.logits_buf
        all_gather_into(self.logits_full, self.logits_buf, grp)          # rank order = vocabulary order
        return self.logits_full

    # ---- on-device greedy chain: replay after replay with nothing in between ------------------------------------
    def start_greedy(self, token: torch.Tensor) -> None:
        off = self.caches[0].offset
        self._gen_next = None
        tok = int(token.reshape(-1)[0].item())
        key = (0x80000000 << 32) | (0xFFFFFFFF - tok)                 # key of (value 0.0, index tok)
        if key >= 1 << 63:
            key -= 1 << 64                                            # same bits as a signed int64
        self.keys.zero_()
        self.keys[0] = key
        if self.tp_vocab is not None:
            self.all_keys.zero_()
            self.all_keys[0] = key
        self.pos.fill_(off - 1)                                       # the embed kernel advances it to `off`
        if self.use_graph:
            self._graph_for(True, self._wants_long(off))              # captures _forward with from_keys / advance_pos set
            self._capture_other_mode(True)

    def step_greedy(self) -> torch.Tensor:
        """One chained greedy step.  Returns ``logits_buf``: the full logits, or -- under a vocabulary-sharded head (tensor
        parallel) -- THIS RANK'S SHARD only (columns tp_vocab[0]:tp_vocab[1]); the chain itself exchanges just the argmax keys.
        Callers that need full logits under TP use ``step()``."""
        off = self.caches[0].offset
        self._check_room(off)
        long = self._wants_long(off)
        if self.use_graph:
            self._graph_for(True, long)[0].replay()
        else:
            self.chain, self.long_mode = True, long
            with torch.no_grad():
                self._forward()
            self.chain, self.long_mode = False, False
        for c in self.caches:
            c.offset = off + 1
        return self.logits_buf


class PromptRunner:
    """Short prompts (2..128 tokens into an EMPTY cache) as ONE replayed HIP graph per length bucket.

    The prompt pass of the reference (gbx_lm/utils.py:312-323: ``model(y[None], cache=prompt_cache)`` then the logits of the last
    position) is ~16 launches per layer: host-bound up to ~100 tokens (Llama-3-8B shapes, measured, scratch/ttft.py: 16 tokens
    7.8 -> 4.1 ms, 128 tokens 8.4 -> 7.5 ms with the graph).  From 256 tokens on the pass is GPU-bound -- the 224 GEMMs at M = 512
    ARE the 14 ms, graph or not (round 2 had that wrong) -- and what helps there is the dense GEMM on resident bf16 images
    (ops.dense_cache_prewarm: 512 tokens 14.2 -> 11.2 ms), so the buckets stop at 128.  A bucket (32, 64, 128 tokens) owns static token ids, static KV staging buffers and the captured forward of the model's OWN body
    over ``bucket`` positions starting at 0: the prompt is padded to the bucket with token 0 (causal attention: positions >= L never
    reach positions < L), the hidden state of position L - 1 is picked by a device-resident index, the head runs on that row only.
    After the replay the first L cache rows of every layer are copied into the caller's cache (its ``offset`` becomes L), so
    everything downstream -- the decode runner, a longer second prompt -- sees exactly what the eager pass would have left.
    Bit-identical to the eager pass (tests/test_prompt_graph_gpu.py): same kernels on the same rows; tile dispatch depends on the
    row count, so the eager comparison runs on the padded bucket as well."""

    BUCKETS = (32, 64, 128)

    def __init__(self, model, bucket: int):
        self.model, self.bucket = model, bucket
        dev = next(model.parameters()).device
        dt = model.model.embed_tokens.weight.dtype
        at = model.model.layers[0].self_attn
        self.ids = torch.zeros((1, bucket), dtype=torch.long, device=dev)
        self.last = torch.zeros((1,), dtype=torch.long, device=dev)
        self.caches = []
        for _ in model.model.layers:
            c = KVCache(step=bucket)
            like = torch.empty((1, at.n_kv_heads, 1, at.head_dim), dtype=dt, device=dev)
            c.reserve(bucket, like, like)
            self.caches.append(c)
        torch.cuda.synchronize(dev)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):                                    # lazy initialisation outside the capture
                self._forward()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = self._forward()

    def _forward(self):
        for c in self.caches:
            c.offset = 0
        hidden = self.model.model(self.ids, None, self.caches)                 # [1, bucket, D]
        h_last = hidden.index_select(1, self.last)                            # [1, 1, D]: position L - 1
        return F.linear(h_last, self.model.head_weight())[:, -1, :]

    def run(self, y: torch.Tensor, prompt_cache: List[KVCache]) -> torch.Tensor:
        L = y.numel()
        self.ids.zero_()
        self.ids[0, :L] = y
        self.last.fill_(L - 1)
        self.graph.replay()
        for dst, src in zip(prompt_cache, self.caches):
            dst.reserve(L, src.keys[:, :, :1], src.values[:, :, :1])
            dst.keys[:, :, :L].copy_(src.keys[:, :, :L])
            dst.values[:, :, :L].copy_(src.values[:, :, :L])
            dst.offset = L
        return self.logits


def prompt_step(model, y: torch.Tensor, prompt_cache) -> Optional[torch.Tensor]:
    """``model.prompt_step`` hook of generate_step: logits [1, V] of the last prompt position through a graph-replayed bucket, or
    None when the call is not a short fresh prompt on the GPU (the caller then runs the eager pass).  GBX_PROMPT_GRAPH=0 disables."""
    import os
    L = y.numel()
    if L < 2 or L > PromptRunner.BUCKETS[-1] or not env.flag("GBX_PROMPT_GRAPH"):
        return None
    c0 = prompt_cache[0] if prompt_cache else None
    if (not isinstance(c0, KVCache) or any(type(c) is not KVCache or c.offset != 0 or c.static_active for c in prompt_cache)
            or not y.is_cuda or getattr(model, "tp_world", 1) > 1 or torch.cuda.is_current_stream_capturing()):
        return None
    if getattr(model, "_prompt_graph_unavailable", False):
        return None
    bucket = next(b for b in PromptRunner.BUCKETS if b >= L)
    # A captured graph bakes in the weights' addresses: key the runners on the storage of the tensors every forward reads first
    # and last (embedding, head, first and last block) and on their in-place version counters, and rebuild on any mismatch
    # (load_weights / .to() / a re-quantise) -- the check ops._DenseCache makes for its images.
    def _sig():
        l0, l1 = model.model.layers[0], model.model.layers[-1]
        ts = (model.model.embed_tokens.weight, model.head_weight(), l0.self_attn.q_proj.qweight, l0.self_attn.q_proj.scales,
              l1.mlp.down_proj.qweight, l1.mlp.down_proj.scales, model.model.norm.weight)
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in ts)
    sig = _sig()
    runners = model.__dict__.setdefault("_prompt_runners", {})
    if runners.get("sig") != sig:
        runners.clear()
        runners["sig"] = sig
    r = runners.get(bucket)
    if r is None:
        try:
            r = runners[bucket] = PromptRunner(model, bucket)
        except Exception as e:      # OOM of the private pool / staging caches, a forward that cannot be captured: the eager pass still works
            import warnings
            warnings.warn(f"graph-replayed prompt pass unavailable ({type(e).__name__}: {e}); using the eager pass")
            model._prompt_graph_unavailable = True
            runners.clear()
            torch.cuda.synchronize()
            return None
    return r.run(y, prompt_cache)


def _rope_at(x: torch.Tensor, rope, pos: torch.Tensor) -> torch.Tensor:
    """RoPE at a device-resident position (graph-safe twin of rope_utils.apply_rope)."""
    inv = rope.inv_freq
    dims = inv.numel() * 2
    ang = (pos.to(torch.float32) * getattr(rope, "scale", 1.0))[:, None] * inv[None, :]
    cos, sin = torch.cos(ang), torch.sin(ang)
    xf = x[..., :dims].to(torch.float32)
    if rope.traditional:
        x1, x2 = xf[..., 0::2], xf[..., 1::2]
        out = torch.stack([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1).flatten(-2)
    else:
        x1, x2 = xf[..., : dims // 2], xf[..., dims // 2:]
        out = torch.cat([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1)
    out = out.to(x.dtype)
    if dims < x.shape[-1]:
        out = torch.cat([out, x[..., dims:]], dim=-1)
    return out


def decode_fault_words(model):
    """Device status words that ``generate_step`` reads next to every token (empty: this model's decode has no in-launch hand-offs)."""
    runner = getattr(model, "_decode_runner", None)
    fw = getattr(runner, "fault_words", None)
    return fw() if fw is not None else []


def decode_fallback(model, code: int, prompt_cache=None, offset: Optional[int] = None) -> None:
    """A fused launch of this model gave up on a hand-off (status ``code``; its stream is NaN from there on): drop the runner and keep
    this model on the five-launch block from now on.  With ``prompt_cache`` / ``offset`` the caches are rewound to ``offset`` cached
    positions and the rows the faulted steps appended behind it are cleared -- they hold NaN, and the attention launches multiply
    speculatively fetched rows beyond the position by a zero weight (0 x NaN = NaN).  The caller then replays the step."""
    import warnings
    if not getattr(model, "_fused_handoff_disabled", False):
        warnings.warn(f"gbx_lm_amd: a fused decode launch timed out on hand-off {code} (another tenant on the GPU, a CU mask, or a "
                      "workgroup placement that broke the one-kv-group-per-XCD bijection); this model decodes on the five-launch block "
                      "from here on and the step is replayed")
    model._fused_handoff_disabled = True
    model._decode_runner = None
    torch.cuda.synchronize()
    if prompt_cache is not None and offset is not None:
        for c in prompt_cache:
            if isinstance(c, KVCache) and c.keys is not None:
                c.keys[:, :, offset:c.offset + 1].zero_()
                c.values[:, :, offset:c.offset + 1].zero_()
            c.offset = offset
        torch.cuda.synchronize()


def decode_step_checked(model, y: torch.Tensor, prompt_cache: List[KVCache]) -> torch.Tensor:
    """``decode_step`` for loops that synchronise on every token anyway (EMINF reads the logits on the host): the fault words are read
    right behind the step and a faulted step is replayed on the five-launch block before its logits reach the caller."""
    off = prompt_cache[0].offset
    logits = decode_step(model, y, prompt_cache)
    for w in decode_fault_words(model):
        code = int(w.item())
        if code:
            decode_fallback(model, code, prompt_cache, off)
            return decode_step(model, y, prompt_cache)
    return logits


def _runner_for(model, prompt_cache: List[KVCache]):
    """The decode runner bound to ``prompt_cache`` (built, or rebuilt when the cache changed or grew); None when the cache is not a
    one-sequence device cache."""
    c0 = prompt_cache[0]
    if not isinstance(c0, KVCache) or c0.keys is None or c0.keys.shape[0] != 1 or not c0.keys.is_cuda:
        return None
    runner: Optional[DecodeRunner] = getattr(model, "_decode_runner", None)
    bound = (runner is not None and runner.caches is prompt_cache and runner.capacity == c0.capacity
             and c0.keys.data_ptr() == getattr(runner, "_kptr", None))
    # A grown cache means new descriptors and new graphs (~10-20 ms).  The reference grows by `step` = 256 positions (cache.py:220-244);
    # with a runner bound to the cache's addresses that was a rebuild every 256 tokens -- and, before the load-time companions were kept on
    # the modules, 0.2-0.4 s each: generate() at half the device rate (scratch/gen_rate2.py).  So: a cache that gets its first runner is
    # grown to four steps up front, and a full one doubles (at most 32 steps = 8192 positions at a time).
    more = 0
    if c0.offset + 1 > c0.capacity:
        more = max(c0.step, min(c0.capacity, 32 * c0.step))
    elif not bound and c0.capacity < 4 * c0.step:
        more = 4 * c0.step - c0.offset
    if more > 0:
        for c in prompt_cache:
            c.reserve(more, c.keys[:, :, :1], c.values[:, :, :1])
        bound = False
    if not bound:
        runner = (FusedDecodeRunner if fused_eligible(model) else DecodeRunner)(model, prompt_cache)
        runner._kptr = prompt_cache[0].keys.data_ptr()
        model._decode_runner = runner
    return runner


def decode_step(model, y: torch.Tensor, prompt_cache: List[KVCache]) -> torch.Tensor:
    """``model.decode_step`` hook used by generate_step for single-token steps: graph replay when possible."""
    runner = _runner_for(model, prompt_cache)
    if runner is None:
        return model(y[None], cache=prompt_cache)[:, -1, :]
    return runner.step(y)


def decode_step_logprobs(model, y: torch.Tensor, prompt_cache: List[KVCache]):
    """``model.decode_step_logprobs`` hook of generate_step's default-sampler loop: (next token int64 [1], logprobs [V], packed
    int64 [4] = (token, hand-off status words)) from ONE graph replay (FusedDecodeRunner.step_gen), or None when that path does not
    apply (the caller then runs decode_step + logprobs_argmax).  ``token`` and ``packed`` are static buffers: valid until the next call."""
    runner = _runner_for(model, prompt_cache)
    if not isinstance(runner, FusedDecodeRunner) or runner.tp_vocab is not None or runner.tp_world > 1 or not runner.use_graph:
        return None
    tok, lp, words = runner.step_gen(y)
    return tok, lp.view(-1).clone(), words
