"""GPU: tensor-parallel fused decode (world_size 2, both ranks on cuda:0, gloo all-reduce of CUDA tensors) must match
the unsharded fused decode.  Exercises shard boundaries through the real HIP kernels (K-split o_proj / down_proj)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out, batch=1, backend="gloo", graph=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if backend == "nccl":                                       # one GPU per rank, RCCL over xGMI (needs >= 2 GPUs)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gbx_lm_amd import synthetic as S, tp
        from gbx_lm_amd.decode import DecodeRunner, FusedDecodeRunner, fused_eligible
        from gbx_lm_amd.models.cache import KVCache
        dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
        cfg = dict(S.TINY, num_hidden_layers=3, hidden_size=512, intermediate_size=1024, num_attention_heads=4,
                   num_key_value_heads=2)                      # head_dim 128; per-rank K: o_proj 256, down_proj 512
        strat = S.make_strategy("S22", 3, 64)
        toks = [5, 9, 300, 17, 2, 44, 101, 7]
        chains = {}

        def run(model):
            at = model.model.layers[0].self_attn
            caches = []
            for _ in model.model.layers:
                c = KVCache(step=64)
                like = torch.empty((batch, at.n_kv_heads, 1, at.head_dim), dtype=torch.bfloat16, device=dev)
                c.reserve(1, like, like)
                caches.append(c)
            if batch > 1:                # the B > 1 runner: batch GEMMs, batched attention, all-reduce after o_proj / the MLP
                r = DecodeRunner(model, caches, use_graph=graph and getattr(model, "tp_world", 1) > 1)
                return [r.step(torch.tensor([(t + 37 * b) % 512 for b in range(batch)], device=dev)).float().cpu().clone() for t in toks]
            r = FusedDecodeRunner(model, caches, use_graph=graph and getattr(model, "tp_world", 1) > 1)   # graph: the collectives are captured
            lg = [r.step(torch.tensor([t], device=dev)).float().cpu().clone() for t in toks]
            # on-device greedy chain (vocabulary-sharded head: 64 argmax keys per rank are exchanged, never the logits)
            r.start_greedy(torch.argmax(lg[-1], -1))
            ids, margins = [], []
            for _ in range(6):
                r.step_greedy()
                ids.append(int(r.tok.item()))
                if getattr(model, "tp_vocab", None) is None:      # unsharded run: how decisive was the NEXT choice (full logits)
                    top2 = r.logits_buf.float().reshape(-1).topk(2).values
                    margins.append(float(top2[0] - top2[1]))
            chains[id(model)] = (ids, margins)
            return lg

        full = S.build_model(cfg, strat, device=dev, seed=9)
        assert fused_eligible(full)
        want = run(full)
        shard = S.build_model(cfg, strat, device=dev, seed=9)
        tp.shard_llama(shard, rank, world)
        assert fused_eligible(shard)
        assert getattr(shard, "tp_vocab", None) is not None and shard.head_weight().shape[0] * world == full.head_weight().shape[0]
        got = run(shard)
        torch.cuda.synchronize()
        tp.p2p_check()                                          # (raises if a one-shot all-reduce gave up on a peer; "p2p off" here)
        out[rank] = max((g - w).abs().max().item() / (w.abs().max().item() + 1e-6) for g, w in zip(got, want))
        noise = 2.0 * max((g - w).abs().max().item() for g, w in zip(got, want))      # what sharding does to a logit, with margin
        # wherever the unsharded top-2 margin exceeds that noise the sharded run must pick the same token (every sequence, every step)
        checked = 0
        for g, w in zip(got, want):
            top2 = w.topk(2, dim=-1)
            decisive = (top2.values[..., 0] - top2.values[..., 1]) > noise
            assert torch.equal(g.argmax(-1)[decisive], top2.indices[..., 0][decisive])
            checked += int(decisive.sum())
        assert checked >= 1, "no decisive step at all: the comparison above checked nothing"
        if batch == 1:
            # the on-device greedy chain through the vocabulary-sharded head: the FULL chain, up to the first step whose unsharded
            # choice was a near-tie (margins[i] is the margin of the choice that produced token i + 1)
            (ids_s, _), (ids_f, margins) = chains[id(shard)], chains[id(full)]
            n = 1
            for mg in margins[:-1]:
                if mg <= noise:
                    break
                n += 1
            n = max(n, 2)                     # (round 3 compared the first two tokens unconditionally)
            assert ids_s[:n] == ids_f[:n], (ids_s, ids_f, margins, noise)
    finally:
        dist.destroy_process_group()


def test_tp2_fused_decode_matches_unsharded():
    assert torch.cuda.is_available()
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for k, v in out.items():
        assert v < 0.05, f"rank {k}: relative max |dlogit| = {v}"


def test_tp2_batched_decode_runner_matches_unsharded():
    """DecodeRunner (B = 4) on two tensor-parallel ranks: the `tp_world > 1` branches of decode.py (_forward: all-reduce after
    o_proj, the sharded MLP with its own all-reduce) against the unsharded runner (k-split partial slices, fused SwiGLU GEMM)"""
    assert torch.cuda.is_available()
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, 4), nprocs=world, join=True)
    assert len(out) == world
    for k, v in out.items():
        assert v < 0.05, f"rank {k}: relative max |dlogit| = {v}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL with N > 1 needs two GPUs (the gpurun pool has one; the driver's 8-GPU node runs this)")
@pytest.mark.parametrize("batch", [1, 4])
def test_tp2_over_rccl_with_the_collectives_inside_the_graph(batch):
    """The same comparison with one GPU per rank and torch.distributed's "nccl" backend (= RCCL over xGMI): the all-reduces after
    o_proj / down_proj and the 512-byte argmax exchange of the vocabulary-sharded head are CAPTURED in the runner's HIP graph and
    replayed -- the path bench.py --gpus N times.  Skips on single-GPU boxes."""
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, batch, "nccl", True), nprocs=world, join=True)
    assert len(out) == world
    for k, v in out.items():
        assert v < 0.05, f"rank {k}: relative max |dlogit| = {v}"
