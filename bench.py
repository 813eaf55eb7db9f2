#!/usr/bin/env python3
"""Headline benchmark: greedy decode, batch 1, Llama-3-8B shapes with the synthetic layer-mix "S22" strategy
(2/3/4-bit, g128, raw 2.22 bpw -- BASELINE.json configs[2]), synthetic packed weights.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python bench.py --gpus N --steps K --warmup W         # spawns its own N ranks (one process per GPU) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            # TP over N GPUs (RCCL all-reduce after o_proj / down_proj)
    python bench.py --gpus 2 --dry-run                    # CPU / gloo rehearsal of the N > 1 control flow (no GPU, no model math)

One "step" = one decoded token = one pass of the hot path (224 QuantizedLinear GEMVs) plus its callers (norms, RoPE,
attention over the KV cache, bf16 lm_head, argmax) with every tensor resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s


def qlinear_bytes(m, M=1):
    """ALGORITHMIC bytes of one QuantizedLinear call: packed words + scales + zeros + x + y."""
    n, k, b, g = m.output_dims, m.input_dims, m.bits, m.group_size
    return n * k * b // 8 + 2 * n * (k // g) * 2 + M * k * 2 + M * n * 2 + (2 * n if m.bias is not None else 0)


def token_bytes(model, context):
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    ql = sum(qlinear_bytes(m) for m in model.modules() if isinstance(m, QuantizedLinear))
    hw = model.head_weight()
    head = hw.numel() * hw.element_size()
    at = model.model.layers[0].self_attn
    kv = 2 * len(model.model.layers) * at.n_kv_heads * at.head_dim * context * 2
    return ql, head, kv


def cpu_baseline(model, cfg, timeout=240.0):
    """The PyTorch-CPU re-expression of the identical GBA arithmetic (unpack -> dequant -> torch.matmul fp32,
    plus the cached-dequant variant and a bf16 head), timed on this box's host cores in a CHILD process (its own thread
    pools, no GPU context): oracle/cpu_baseline.py.  The C port of the oracle rides along as a second field.  Reported
    baseline, never the product path."""
    import subprocess
    at = model.model.layers[0].self_attn
    spec = {"hidden": cfg["hidden_size"], "inter": cfg["intermediate_size"], "vocab": cfg["vocab_size"],
            "kv_dim": at.n_kv_heads * at.head_dim, "group": model.model.layers[0].mlp.gate_proj.group_size,
            "layer_bits": [blk.mlp.gate_proj.bits for blk in model.model.layers]}
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(os.cpu_count() or 1)
    env["OMP_WAIT_POLICY"] = "passive"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), json.dumps(spec)],
                         env=env, capture_output=True, text=True, timeout=timeout)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def kernel_leg(model, runner=None, reps=20):
    """Dominant hot-path kernel timed live with HIP events on the launch stream: the fused
    [RMSNorm + gate_proj + up_proj + SwiGLU] GEMV launch of the narrowest-bit layers -- exactly the launch the decode
    graph replays (20 % of a token) -- cycled over every such layer's own weights (>> 256 MiB, so the Infinity Cache
    cannot serve them), IN ITS CHAIN POSITION: a graph of (predecessor launch, gate|up launch) pairs per layer is replayed, then the
    same graph without the gate|up launches; the launch's duration is the difference per pair (boundary included).  The
    predecessor is what the decode runner launches in front of it: the fused attention launch (gbx_dattn_forward, round 5: the
    runner's OWN descriptors are replayed, the gate|up launch then forms x from the fixed-point o_proj sums) or, in the
    five-launch block, the o_proj launch.  (Replaying the 33 MB launch back to back with itself reads 10-15 % longer than
    rocprofv3 shows it inside the token -- every launch then starts on caches its predecessor has just swept -- and is
    reported as `us_back_to_back`.)  The kernel's name comes from the library (gbx_dgemv_kernel_id), not from a re-derivation here.
    Algorithmic bytes per launch = both packed matrices + their scales/zeros + x + norm weight + out
    (+ the 8-byte fixed-point sums the prologue reads in the three-launch block)."""
    from gbx_lm_amd import env, ops
    from gbx_lm_amd import _lib as _l
    layers = list(model.model.layers)
    bits = min(blk.mlp.gate_proj.bits for blk in layers)
    idx = [i for i, blk in enumerate(layers) if blk.mlp.gate_proj.bits == bits and blk.mlp.up_proj.bits == bits]
    layers = [layers[i] for i in idx]
    g0 = layers[0].mlp.gate_proj
    dev = g0.qweight.device
    dt = g0.scales.dtype
    o0 = layers[0].self_attn.o_proj
    xo = torch.randn(o0.input_dims, device=dev).to(dt) * 0.1        # attention output feeding o_proj
    x = torch.randn(g0.input_dims, device=dev).to(dt)               # the residual stream o_proj updates and gate|up normalises
    x0 = x.clone()
    act = torch.empty(g0.output_dims, device=dev, dtype=dt)
    use_dg = env.flag("GBX_DGEMV")
    use_ssq = env.flag("GBX_DG_SSQ")
    pl = ops.dgemv_plan(g0.output_dims, g0.input_dims, g0.bits, g0.group_size, _l.GBX_DG_SWIGLU, True) if use_dg else None
    plo = ops.dgemv_plan(o0.output_dims, o0.input_dims, o0.bits, o0.group_size, _l.GBX_DG_RESIDUAL) if use_dg else None
    keep = []
    acc_bytes = 0
    fused_mlp = False
    pred = "o_proj"
    da = getattr(runner, "da", None)
    db = getattr(runner, "db", None)
    if db is not None and all(db[i] is not None for i in idx) and runner.caches[0].offset <= runner.db_max_ctx:
        return _block_leg(model, runner, idx, reps)
    if da is not None and all(da[i] is not None for i in idx):
        # the three-launch block: the runner's own (fused attention, gate|up) descriptors -- its weights, caches and buffers
        o_descs = [da[i][0] for i in idx]
        g_descs = [da[i][1] for i in idx]
        x, x0 = runner.hb, runner.hb.clone()
        acc_bytes = 8 * g0.input_dims
        pred = "fused attention"
        if isinstance(g_descs[0], _l.DmlpDesc):
            # two-launch block: the dominant launch is the fused MLP launch (gate|up + SwiGLU + down_proj in one kernel)
            dn = layers[0].mlp.down_proj
            fused_mlp = True
            kname = ("gbx::dmlp_kernel<2> (FP4 x FP8 matrix cores; gate|up + SwiGLU + XCD-local hand-off + down_proj)" if bits == 2 else
                     f"gbx::dmlpv_kernel<{bits}, 2, true> (vector-ALU dots; gate|up + SwiGLU + XCD-local hand-off + down_proj from LDS)")
            acc_bytes += (dn.output_dims * dn.input_dims * dn.bits // 8 + 2 * dn.output_dims * (dn.input_dims // dn.group_size) * 2
                          + 8 * dn.output_dims - g0.output_dims * 2)       # + down_proj's matrix, scales / zeros, its fixed-point sums; the activations never leave the chip
        else:
            kname = "gbx::" + ops.dgemv_kernel_name(g_descs[0])
        launch = lambda d, dv: (ops.dattn(d, dv) if isinstance(d, _l.DattnDesc) else
                                (ops.dmlp(d, dv) if isinstance(d, _l.DmlpDesc) else ops.dgemv(d, dv)))
    elif pl is not None and plo is not None:
        # exactly the launches the decode runner replays: gbx_dgemv_forward, the RMSNorm statistic carried from o_proj to gate|up
        ssq = torch.zeros(_l.GBX_DG_SSQ_SLOTS, dtype=torch.float32, device=dev) if (use_ssq and plo[1] > 0) else None
        o_descs, g_descs = [], []
        for blk in layers:
            po, pg = ops.dgemv_pack(blk.self_attn.o_proj, plo[0]), ops.dgemv_pack(blk.mlp.gate_proj, pl[0], blk.mlp.up_proj)
            keep += [po, pg]
            o_descs.append(ops.make_dgemv_desc(xo, blk.self_attn.o_proj, x, po, plo[0], residual=x, ssq_out=ssq))
            g_descs.append(ops.make_dgemv_desc(x, blk.mlp.gate_proj, act, pg, pl[0], mod_up=blk.mlp.up_proj,
                                               norm_weight=blk.post_attention_layernorm.weight.to(dt),
                                               eps=blk.post_attention_layernorm.eps, ssq_in=ssq))
        kname = "gbx::" + ops.dgemv_kernel_name(g_descs[0])
        launch = ops.dgemv
    else:
        o_descs, g_descs = [], []
        for blk in layers:
            po, pg = ops.pack_scales_zeros(blk.self_attn.o_proj), ops.pack_scales_zeros(blk.mlp.gate_proj, blk.mlp.up_proj)
            keep += [po, pg]
            o_descs.append(ops.make_fused_desc(xo, (blk.self_attn.o_proj,), (x,), residual=x, packed=po))
            g_descs.append(ops.make_fused_desc(x, (blk.mlp.gate_proj, blk.mlp.up_proj), (act,),
                                               norm_weight=blk.post_attention_layernorm.weight.to(dt),
                                               eps=blk.post_attention_layernorm.eps, swiglu=True, packed=pg))
        launch, kname = ops.qlinear_fused, f"gbx::gemv_kernel<{bits},...>"

    def timed(seq):
        for d in seq:
            launch(d, dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for d in seq:
                launch(d, dev)
        g.replay()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            x.copy_(x0)                                              # the residual stream grows by o_proj's output every launch
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        return best

    pairs = [d for od, gd in zip(o_descs, g_descs) for d in (od, gd)]
    us_pairs, us_o, us_b2b = timed(pairs), timed(o_descs), timed(g_descs)
    us = (us_pairs - us_o) / len(g_descs)
    descs = g_descs
    n, k, grp = g0.output_dims, g0.input_dims, g0.group_size
    nbytes = 2 * (n * k * bits // 8 + 2 * n * (k // grp) * 2) + k * 2 + k * 2 + n * 2 + acc_bytes
    achieved = nbytes / us / 1e3
    traffic = None            # PMC-derived HBM bytes per launch for this exact kernel/shape, if a profile run recorded it
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(f"{'dmlp' if fused_mlp else 'gateup'}_b{bits}_N{n}_K{k}_g{grp}")
            traffic = rec["traffic_bytes_per_launch"] if rec else None
    except (OSError, ValueError, KeyError):
        traffic = None
    # reference points measured on the same chip (profiles/r04_stream_floor.txt, scratch/probe4/probe_stream.hip): a launch that only
    # STREAMS these bytes (no prologue, no arithmetic), replayed the same way, takes 5.7-6.5 us -- the floor of this launch
    floor = {"pure_stream_us": 5.7, "source": "profiles/r04_stream_floor.txt (29.4 MB of 1 KB rows, 16-wave workgroups, boundary included)"}
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "floor": floor,
            "kernel": (f"{kname} residual+RMSNorm+gate_proj+up_proj+SwiGLU+down_proj, 2 x [N={n}, K={k}] + [N={k}, K={n}] g={grp}" if fused_mlp else
                       f"{kname} fused RMSNorm+gate_proj+up_proj+SwiGLU, 2 x [N={n}, K={k}] g={grp}"),
            "bytes_per_launch": nbytes, "us_per_launch": us, "us_back_to_back": us_b2b / len(descs),
            "method": f"HIP events: graph of ({pred}, {'fused MLP' if fused_mlp else 'gate|up'}) pairs minus the same graph without the second launch, per pair",
            "launches_timed": reps * len(descs),
            "working_set_bytes": nbytes * len(descs)}


def _block_leg(model, runner, idx, reps):
    """Round 6: the dominant launch is the WHOLE 2-bit block (gbx::dblock_kernel: norm -> q|k|v -> attention -> o_proj -> chip-wide release ->
    norm -> gate|up from LDS -> down_proj).  Timed in its chain position: the runner's own one-launch descriptors of the 2-bit layers, in
    layer order, as one graph (every launch's predecessor is the previous block's launch, as in the token), HIP events on the launch
    stream; one tiny accumulator-clearing launch closes the chain so that every replay starts from zero sums.  The duration per launch
    includes its boundary.  Algorithmic bytes per launch = the seven packed matrices + their scales / zeros + the cached K / V rows the
    attention reads + the stream / norm-weight vectors + the two fixed-point accumulators (written and read once each)."""
    from gbx_lm_amd import ops
    dev = runner.hb.device
    layers = [model.model.layers[i] for i in idx]
    descs = [runner.db[i] for i in idx]
    ctx = runner.caches[0].offset
    accs = [t for t in (runner.acc_d[0], runner.acc_d[1], runner.acc_o[0], runner.acc_o[1])]
    tmp = torch.zeros_like(runner.hb)

    def chain():
        for d in descs:
            ops.dblock(d, dev)
        ops.acc_residual(runner.pb, accs[0], tmp, zero=accs[1:])

    hb0 = runner.hb.clone()
    chain()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    g.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    runner.dattn_check()
    runner.hb.copy_(hb0)
    us = best / len(descs)
    blk = layers[0]
    mods = (blk.self_attn.q_proj, blk.self_attn.k_proj, blk.self_attn.v_proj, blk.self_attn.o_proj, blk.mlp.gate_proj, blk.mlp.up_proj, blk.mlp.down_proj)
    D = runner.hb.numel()
    wbytes = sum(m.qweight.numel() * 4 + m.scales.numel() * 2 + m.zeros.numel() * 2 for m in mods)
    kv = 2 * (ctx + 1) * blk.self_attn.n_kv_heads * blk.self_attn.head_dim * 2
    nbytes = wbytes + kv + 4 * D * 2 + 2 * 2 * (D + 1) * 8
    achieved = nbytes / us / 1e3
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get("dblock_b2_D4096_I14336")
            traffic = rec["traffic_bytes_per_launch"] if rec else None
    except (OSError, ValueError, KeyError):
        traffic = None
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "kernel": "gbx::dblock_kernel (one launch per 2-bit decoder block: RMSNorm + q|k|v + RoPE/append + attention + o_proj + chip-wide release + "
                      "RMSNorm + gate|up from LDS + SwiGLU + down_proj; FP4 x FP8 matrix cores), Llama-3-8B shapes, g=128",
            "bytes_per_launch": nbytes, "us_per_launch": us, "context": ctx,
            "note": "round 6: the dominant launch is the WHOLE 2-bit block (attention latency chain + MLP), 65 % of the token; rounds 4-5 quoted the MLP / gate|up "
                    "launch alone (0.40-0.43 of 8 TB/s: GBX_DBLOCK=0 reproduces it).  Over the SAME bytes the two launches of round 5 take 27.6 us stamped "
                    "(0.28 of 8 TB/s) against 22.8 us for this launch (profiles/r06_layer_timeline_two_launch.txt / _dblock.txt; 947-962 vs 1031-1050 tok/s "
                    "on one box, profiles/r06_dblock_ab.txt)",
            "same_scope_two_launch": {"us_per_block": 27.6, "frac": round(nbytes / 27.6 / 1e3 / HBM_PEAK_GBPS, 3),
                                      "source": "profiles/r06_layer_timeline_two_launch.txt (GBX_DBLOCK=0: gbx_dattn_forward + gbx_dmlp_forward, stamped spans + boundaries)"},
            "method": "HIP events: one graph of the runner's own one-launch descriptors of every 2-bit layer in layer order (+ one accumulator-clearing "
                      "launch), per launch, boundary included",
            "launches_timed": reps * len(descs), "working_set_bytes": nbytes * len(descs)}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process has
    touched the GPU, forward rank 0's JSON line, exit with the worst return code.  Never re-executes a process."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rc = procs[0].returncode
    for pr in procs[1:]:
        try:
            pr.wait(timeout=600 if rc == 0 else 20)
        except subprocess.TimeoutExpired:
            pr.kill()                                   # exactly the child we started
        rc = rc or pr.returncode or (1 if pr.returncode is None else 0)
    sys.stdout.write(out0)
    sys.stdout.flush()
    raise SystemExit(rc)


def dry_run(args, world, rank):
    """CPU rehearsal of the N > 1 control flow: gloo rendezvous, tp.shard_llama of the tiny model, the path's exchange
    pattern (one all-reduce of [B, D] after o_proj and after down_proj per layer) timed with the same barrier / max-over-
    ranks protocol, ONE JSON line from rank 0.  No QuantizedLinear math runs here (that needs the HIP extension)."""
    from gbx_lm_amd import env as genv, synthetic as S, tp
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    # tiny shapes that still shard 8 ways (K / 8 a multiple of 128 values, 8 KV heads)
    cfg = dict(S.TINY, hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=8)
    model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"]), device="cpu", seed=0)
    if world > 1:
        tp.shard_llama(model, rank, world)
    h = torch.zeros((args.batch, cfg["hidden_size"]), dtype=torch.float32)

    def step():
        for _ in model.model.layers:
            for _ in range(2):
                if world > 1:
                    tp.all_reduce_(h)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ms = (time.perf_counter() - t0) * 1e3
    if world > 1:
        dist.barrier()
        t = torch.tensor([ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    if rank == 0:
        q = model.model.layers[0].self_attn.q_proj
        print(json.dumps({"metric": "dry_run_exchange_steps_per_s", "value": args.steps * 1e3 / max(ms, 1e-9), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms / args.steps,
                          "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
                          "dtype": "none", "data": "synthetic", "dry_run": True, "rccl_ranks": 0, "backend": "gloo",
                          "config": {"workload": "tiny Llama, TP shard + all-reduce pattern only (CPU rehearsal)",
                                     "parallelism": f"tp{world}", "q_rows_per_rank": q.output_dims}}))
    if world > 1:
        dist.destroy_process_group()


def time_decode(runner, steps, warmup, batch, world=1, dev=None):
    """EXACTLY `steps` greedy decode steps between a barrier + synchronize on both sides; HIP events on the launch stream
    (N = 1) / max(events, host clock) then MAX over ranks (N > 1).  Returns ms for the `steps` steps."""
    for _ in range(warmup):
        runner.step_greedy()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_host0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        runner.step_greedy()
    e1.record()
    torch.cuda.synchronize()
    t_host = time.perf_counter() - t_host0
    if world > 1:
        dist.barrier()
    ms = max(e0.elapsed_time(e1), t_host * 1e3) if world > 1 else e0.elapsed_time(e1)
    if world > 1:
        tms = torch.tensor([ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tms, op=dist.ReduceOp.MAX)
        ms = float(tms.item())
    return ms


def make_caches(model, batch, cap, dev):
    from gbx_lm_amd.models.cache import KVCache
    at = model.model.layers[0].self_attn
    caches = []
    for _ in model.model.layers:
        c = KVCache(step=cap)
        like = torch.empty((batch, at.n_kv_heads, 1, at.head_dim), dtype=torch.bfloat16, device=dev)
        c.reserve(1, like, like)
        caches.append(c)
    return caches


def decode_leg(model, cfg, dev, batch, context, steps, warmup, world=1, use_graph=True):
    """A secondary decode point of the metric (same protocol as the headline leg; with world > 1 every rank runs it on its
    shard and the time is the maximum over ranks)."""
    from gbx_lm_amd.decode import DecodeRunner, FusedDecodeRunner, fused_eligible
    cap = ((context + warmup + steps + 2 + 255) // 256) * 256
    caches = make_caches(model, batch, cap, dev)
    Runner = FusedDecodeRunner if (fused_eligible(model) and batch == 1) else DecodeRunner
    runner = Runner(model, caches, use_graph=use_graph)
    full = None
    for t in range(context):
        full = runner.step(torch.full((batch,), t % cfg["vocab_size"], device=dev, dtype=torch.long))
    # seed the greedy chain from the GATHERED logits step() returns (under a vocabulary-sharded head runner.logits is this rank's
    # shard: an argmax over it would start every rank on a different token id)
    runner.start_greedy(torch.argmax(full, -1))
    ms = time_decode(runner, steps, warmup, batch, world, dev) / steps
    if hasattr(runner, "dattn_check"):
        runner.dattn_check()
    runner.norm_check()
    if world > 1:
        from gbx_lm_amd import tp as _tp
        _tp.p2p_check()
    ql_b, head_b, kv_b = token_bytes(model, context + warmup + steps // 2)
    nbytes = ql_b + head_b + kv_b * batch
    return {"tok_s": batch * 1e3 / ms, "ms_per_step": ms, "steps": steps, "context": context, "batch": batch,
            "bytes_per_step": nbytes, "achieved_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBPS,
            "decode_block": Runner.__name__, "parallelism": f"tp{world}"}


def prefill_leg(model, cfg, dev, B=8, T=4096):
    """BASELINE.json configs[3]: prefill of B x T tokens through the model's own forward (chunks of 2048 like
    generate_step), then the QuantizedLinear calls alone at the same M (MFMA-bound: fraction of the 2.5 PFLOP/s dense
    bf16 peak from the algorithmic flops 2 M N K).

    Every number comes in two states of ``ops._DenseCache`` (the resident bf16 images of the packed weights, 2 bytes per weight,
    that a long call leaves behind): COLD = cache emptied inside the timed region, every QuantizedLinear call dequantises its
    weight first (a dequant + matmul, what a first request pays); WARM = images resident, the timed calls are dense bf16 GEMMs on
    pre-expanded weights (what the second chunk of a prompt and every later request pay)."""
    from gbx_lm_amd import ops
    from gbx_lm_amd.models.cache import make_prompt_cache
    toks = (torch.arange(B * T, device=dev) % cfg["vocab_size"]).view(B, T)

    def run():
        cache = make_prompt_cache(model)
        with torch.no_grad():
            for s0 in range(0, T, 2048):
                model.model(toks[:, s0:s0 + 2048], cache=cache)

    def timed(fn, cold):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if cold:
            ops.dense_cache_clear()                 # inside the timed region: the images are rebuilt under the clock
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    run()                                            # lazy initialisation, allocator warm-up
    ms_cold = timed(run, True)
    image_bytes = ops.dense_cache_bytes()
    ms_warm = timed(run, False)
    mods = [m for blk in model.model.layers for m in (blk.self_attn.q_proj, blk.self_attn.k_proj, blk.self_attn.v_proj,
                                                      blk.self_attn.o_proj, blk.mlp.gate_proj, blk.mlp.up_proj, blk.mlp.down_proj)]
    flops = 2.0 * B * T * sum(m.input_dims * m.output_dims for m in mods)
    x = {k: torch.randn(B * 2048, k, device=dev).to(torch.bfloat16) for k in {m.input_dims for m in mods}}

    def qlinear_only():
        with torch.no_grad():
            for _ in range(T // 2048):
                for m in mods:
                    m(x[m.input_dims])
    tg_cold = timed(qlinear_only, True)
    tg_warm = timed(qlinear_only, False)
    return {"tok_s": B * T / ms_warm * 1e3, "ms": ms_warm, "tok_s_cold": B * T / ms_cold * 1e3, "ms_cold": ms_cold,
            "batch": B, "seq_len": T, "dense_image_bytes": image_bytes,
            "qlinear_flops": flops,
            "qlinear_ms": tg_warm, "qlinear_TFLOPs": flops / tg_warm / 1e9, "frac_of_2500_TFLOPs_dense_bf16": flops / tg_warm / 1e9 / 2500.0,
            "qlinear_ms_cold": tg_cold, "qlinear_TFLOPs_cold": flops / tg_cold / 1e9,
            "frac_of_2500_TFLOPs_dense_bf16_cold": flops / tg_cold / 1e9 / 2500.0,
            "note": "ms / tok_s: whole forward incl. gbx_attn_prefill (HIP causal attention) and glue (RoPE, norms, cache writes). qlinear_*: "
                    "the 224 QuantizedLinear calls of the model alone at M = B*2048 rows, two chunks. *_cold: ops.dense_cache_clear() inside "
                    "the timed region, so every call dequantises its packed weight once (dequant + matmul); without the suffix the "
                    "dequantised bf16 images (dense_image_bytes of HBM, 2 bytes per weight) are resident and the timed calls are dense "
                    "bf16 GEMMs on pre-expanded weights -- no dequantisation runs in that loop"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512, help="timed decode steps (default: > 0.5 s of GPU time, so that a coarse GPU-busy sampler sees the run)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--context", type=int, default=128, help="synthetic prompt length already in the KV cache")
    ap.add_argument("--strategy", default="S22", choices=["S22", "S40", "S20", "S30"])
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3.2-3b", "tiny"])
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded synchronously (1 = the headline workload; "
                    "32 = BASELINE.json configs[4], through the generic graph-replayed runner)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-leg", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs (bpw-4.0 decode, batch-32 decode, prefill)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo rehearsal of the multi-rank control flow")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                                     # never returns; nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the QuantizedLinear path has no CPU fallback)")
    t_start = time.perf_counter()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank that dies must not leave the others waiting for the default 10 minutes
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=240))

    from gbx_lm_amd import synthetic as S, tp
    from gbx_lm_amd.decode import DecodeRunner, FusedDecodeRunner, fused_eligible

    cfg = {"llama3-8b": S.LLAMA3_8B, "llama3.2-3b": S.LLAMA32_3B, "tiny": S.TINY}[args.model]
    strat = S.make_strategy(args.strategy, cfg["num_hidden_layers"], 128)
    model = S.build_model(cfg, strat, device=dev, seed=0)          # same seed on every rank -> identical weights
    tp_allreduce = "none"
    if world > 1:
        tp.shard_llama(model, rank, world)
        tp_allreduce = "rccl"
        if genv.flag("GBX_TP_P2P"):
            # opt-in: one-shot push all-reduce through hipIpc inboxes (gbx_p2p_allreduce) instead of RCCL for the [B, D] exchanges
            if tp.enable_p2p(rank, world, cfg["hidden_size"] * 32, dev) is not None:
                tp_allreduce = "p2p one-shot (gbx_p2p_allreduce)"
            else:
                tp_allreduce = f"rccl (p2p unavailable: {tp.p2p_unavailable_reason})"
    torch.cuda.synchronize()

    total = args.context + args.warmup + args.steps + 2
    cap = ((total + 255) // 256) * 256
    caches = make_caches(model, args.batch, cap, dev)

    # N > 1: the RCCL all-reduces are captured into the same HIP graph; GBX_TP_GRAPH=0 forces eager launches, and a capture
    # that raises (before any collective has been enqueued for real) falls back to them on every rank alike
    use_graph = not args.no_graph and (world == 1 or genv.flag("GBX_TP_GRAPH"))
    graph_note = "hipGraph replay"
    Runner = FusedDecodeRunner if (fused_eligible(model) and args.batch == 1) else DecodeRunner
    try:
        runner = Runner(model, caches, use_graph=use_graph)
    except Exception as e:                                          # e.g. a collective that refuses capture
        if not use_graph:
            raise
        runner = Runner(model, caches, use_graph=False)
        graph_note = f"eager launches (graph capture failed: {type(e).__name__})"
    if not use_graph:
        graph_note = "eager launches"

    vocab = cfg["vocab_size"]
    # synthetic fixed-shape prompt: token ids arange(L) mod vocab, fed through the decode path to fill the KV cache
    full = None
    for t in range(args.context):
        full = runner.step(torch.full((args.batch,), t % vocab, device=dev, dtype=torch.long))
    tok = torch.argmax(full, -1)                                    # step() returns the gathered logits under a sharded head
    tp_selfcheck = None
    if world > 1:
        # N > 1 self-check (so that the first real SCALE line is interpretable): the process group really has N ranks, every rank
        # starts the greedy chain on the SAME token, and what the 64 all-reduces of a token cost on their own
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
        toks = [torch.zeros_like(tok) for _ in range(world)]
        dist.all_gather(toks, tok)
        if not all(torch.equal(t, toks[0]) for t in toks):
            raise RuntimeError(f"rank {rank}: the ranks disagree on the first greedy token: {[int(t.reshape(-1)[0]) for t in toks]}")
        buf = torch.zeros((cfg["hidden_size"],), dtype=torch.bfloat16, device=dev)
        for _ in range(8):
            tp.all_reduce_(buf)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(64):
            tp.all_reduce_(buf)
        e1.record()
        torch.cuda.synchronize()
        tp_selfcheck = {"ranks": dist.get_world_size(), "first_token_agrees": True,
                        "allreduce_us": e0.elapsed_time(e1) * 1e3 / 64, "allreduce_bytes": buf.numel() * 2,
                        "note": "64 eager all-reduces of one [hidden] bf16 vector, back to back on rank 0's stream (a token has 64)"}
    runner.start_greedy(tok)                                        # greedy chain: argmax + next-token feed stay on the device

    ms = time_decode(runner, args.steps, args.warmup, args.batch, world, dev)
    if hasattr(runner, "dattn_check"):
        runner.dattn_check()            # a fused attention launch whose hand-off timed out voids the run (raises; off the timed path)
    runner.norm_check()
    if world > 1:
        tp_allreduce += "; " + tp.p2p_check()      # a one-shot all-reduce that gave up on a peer voids the run (raises -> non-zero exit)
    launches = [d is not None for d in getattr(runner, "da", [])]
    n2 = sum(1 for d in getattr(runner, "da", []) if d is not None and len(d) == 2)
    block_note = ((f"{n2} blocks on 2 launches (gbx_dattn_forward, gbx_dmlp_forward), {len(launches) - n2} on 3 (gbx_dattn_forward, gate|up, down)"
                   if n2 else "3 launches per block (fused attention launch gbx_dattn_forward, gate|up, down)") if launches and all(launches) else
                  (f"{sum(launches)} of {len(launches)} blocks on 3 launches, the rest on 5" if any(launches) else "5 launches per block"))
    ms_per_step = ms / args.steps
    tok_s = args.batch * 1e3 / ms_per_step

    ql_b, head_b, kv_b = token_bytes(model, args.context + args.warmup + args.steps // 2)
    per_gpu_bytes = ql_b + head_b + kv_b
    e2e_gbps = per_gpu_bytes * (tok_s / args.batch) / 1e9           # per GPU and step (each rank streams its own shard once per step)

    roof = None
    cpu = None
    extra = None
    if rank == 0 and not args.no_kernel_leg:
        roof = kernel_leg(model, runner)
    if rank == 0 and world == 1 and not args.no_extra and args.model == "llama3-8b" and args.batch == 1:
        # the rest of the metric (BASELINE.json: "bpw-2.2 & 4.0", configs[3], configs[4] at TP 1) in the same line, bounded
        extra = {}
        legs = []

        def generate_leg():
            # the PRODUCT loop with the host in it (gbx_lm.generate_step, utils.py:217-338): one token id + the hand-off status words cross
            # PCIe per token, step n + 1 is enqueued before token n is read.  Never `value`; stated beside it.
            from gbx_lm_amd.utils import generate_step
            prompt = (torch.arange(args.context) * 7 + 3) % 32000
            n_tok, best = 160, 0.0
            for _ in range(2):                                      # (the first call builds the runner and captures its graphs)
                t0 = None
                for i, _tok in enumerate(generate_step(prompt, model, max_tokens=n_tok)):
                    if i == 7:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                torch.cuda.synchronize()
                best = max(best, (n_tok - 8) / (time.perf_counter() - t0))
            model._decode_runner = None
            return {"tok_s": best, "tokens": n_tok - 8, "context": args.context, "host_in_loop": True,
                    "note": "generate_step, default sampler: one graph replay per token (blocks + head + logprobs + argmax, token fed back on the "
                            "device), logprobs cloned per token, (token, status words) in one D2H copy; contexts grow from `context` on"}
        legs.append(("generate_step_" + args.strategy, generate_leg))
        if args.strategy == "S22":
            legs.append(("b32_decode_S22", lambda: decode_leg(model, cfg, dev, 32, 128, 16, 2)))

        def s40_legs():
            m40 = S.build_model(cfg, S.make_strategy("S40", cfg["num_hidden_layers"], 128), device=dev, seed=0)
            out = {"decode": decode_leg(m40, cfg, dev, 1, args.context, args.steps, args.warmup)}
            try:
                out["prefill_B8_T4096"] = prefill_leg(m40, cfg, dev)
            except Exception as e:
                out["prefill_B8_T4096"] = {"failed": f"{type(e).__name__}: {e}"[:200]}
            return out
        if args.strategy != "S40":
            legs.append(("S40", s40_legs))

        def cfg2_leg():                                             # BASELINE.json configs[1]: Llama-3.2-3B shapes, all 4-bit, batch 1
            c3 = S.LLAMA32_3B
            m3 = S.build_model(c3, S.make_strategy("S40", c3["num_hidden_layers"], 128), device=dev, seed=0)
            return decode_leg(m3, c3, dev, 1, args.context, args.steps, args.warmup)
        legs.append(("llama3.2-3b_S40_decode", cfg2_leg))
        for name, fn in legs:
            if time.perf_counter() - t_start > 240.0:
                extra[name] = {"skipped": "time budget"}
                continue
            try:
                extra[name] = fn()
            except Exception as e:
                extra[name] = {"failed": f"{type(e).__name__}: {e}"[:200]}
            torch.cuda.empty_cache()
    if world > 1 and not args.no_extra and args.model == "llama3-8b" and args.batch == 1 and args.strategy == "S22":
        # BASELINE.json configs[4] on the N ranks of this run: batch-32 greedy decode, row / column-sharded QuantizedLinear, one
        # all-reduce after o_proj and after down_proj (every rank takes part: the collectives are in the step)
        try:
            leg = decode_leg(model, cfg, dev, 32, 128, 16, 2, world=world, use_graph=use_graph)
        except Exception as e:
            leg = {"failed": f"{type(e).__name__}: {e}"[:200]}
        if rank == 0:
            extra = {"b32_decode_S22": leg}
    if rank == 0 and world > 1:
        extra = dict(extra or {}, tp_selfcheck=tp_selfcheck, allreduce_us=tp_selfcheck["allreduce_us"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(model, cfg)
        except Exception as e:                                      # never let the baseline leg sink the bench line
            cpu = {"value": None, "unit": "tok/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if world > 1:
        dist.barrier()

    if rank == 0:
        out = {
            "metric": "decode_tok_per_s", "value": tok_s, "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "rccl_ranks": world if world > 1 else 0,
            "config": {"workload": f"{args.model} layer-mix {args.strategy} greedy decode batch={args.batch} "
                                   f"context={args.context} (BASELINE.json configs[{2 if args.batch == 1 else 4}])",
                       "global_batch": args.batch, "seq_len": args.context, "parallelism": f"tp{world}",
                       "launch": graph_note, "decode_block": Runner.__name__, "block_launches": block_note, "bits_mix": args.strategy, "group_size": 128,
                       "tp_allreduce": tp_allreduce, "tp_head": "vocabulary-sharded" if getattr(model, "tp_vocab", None) else "full"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "e2e": {"bytes_per_token_per_gpu": per_gpu_bytes, "qlinear_bytes": ql_b, "lm_head_bytes": head_b,
                    "kv_bytes": kv_b, "achieved_GBps_per_gpu": e2e_gbps, "frac_of_8TBps": e2e_gbps / HBM_PEAK_GBPS,
                    "frac_of_measured_copy_6290": e2e_gbps / 6290.0},
            "extra": extra,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
