"""Tensor-parallel sharding (world_size 2, gloo, CPU): row/column shards + one all-reduce per K-split layer must
reproduce the unsharded model.  The QuantizedLinear op is the oracle here (test-only), so this covers the host
logic of the N > 1 path: shard boundaries on packed words / groups, head split, where the collective sits."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import oracle_matmul
        from gbx_lm_amd import ops, synthetic as S, tp
        ops.quantized_matmul = oracle_matmul
        cfg = dict(S.TINY)
        strat = S.make_strategy("S22", 2)
        strat["model.layers.0"]["o_proj"] = {"bits": [3], "group_size": {"3": 32}}
        strat["model.layers.1"]["down_proj"] = {"bits": [2], "group_size": {"2": 64}}
        strat["model.layers.1"]["q_proj"] = {"bits": [2], "group_size": {"2": 32}}
        full = S.build_model(cfg, strat, device="cpu", seed=21)
        ids = torch.tensor([[3, 9, 27, 81, 243, 11]])
        with torch.no_grad():
            want = full(ids).float()
        shard = S.build_model(cfg, strat, device="cpu", seed=21)
        tp.shard_llama(shard, rank, world)
        q = shard.model.layers[0].self_attn.q_proj
        assert q.output_dims == 256 // world and q.qweight.shape[0] == 256 // world
        d = shard.model.layers[1].mlp.down_proj
        assert d.input_dims == 512 // world and d.qweight.shape == (256, 512 // world * d.bits // 32)
        assert d.scales.shape == (256, 512 // world // d.group_size)
        with torch.no_grad():
            got = shard(ids).float()
        err = (got - want).abs().max().item()
        out[rank] = err
        # decode with KV cache under TP: cache holds this rank's KV heads only
        from gbx_lm_amd.models.cache import make_prompt_cache
        c = make_prompt_cache(shard)
        with torch.no_grad():
            l0 = shard(ids[:, :4], cache=c)
            l1 = shard(ids[:, 4:], cache=c)
        assert c[0].keys.shape[1] == cfg["num_key_value_heads"] // world
        out[rank + world] = (l1.float() - want[:, 4:]).abs().max().item()
    finally:
        dist.destroy_process_group()


def test_tp2_matches_unsharded():
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == 2 * world
    # partial sums are rounded to bf16 before the all-reduce: a few bf16 ulps of the logits' scale
    for k, v in out.items():
        assert v < 0.1, f"rank/check {k}: max |dlogit| = {v}"


def test_shard_validation():
    sys.path.insert(0, ROOT)
    from gbx_lm_amd import tp
    from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
    m = QuantizedLinear(256, 96, False, 64, 4)
    with pytest.raises(ValueError):
        tp.shard_rows(m, 0, 5)
    with pytest.raises(ValueError):
        tp.shard_cols(m, 0, 4)          # 256 / 4 = 64 values: not a multiple of 128
    m2 = QuantizedLinear(1024, 96, True, 128, 3)
    tp.shard_cols(m2, 1, 4)
    assert m2.input_dims == 256 and m2.qweight.shape == (96, 24) and m2.scales.shape == (96, 2)
    assert torch.count_nonzero(m2.bias) == 0
