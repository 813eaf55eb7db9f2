"""gbx_p2p_allreduce (csrc/p2p_allreduce.hip): the one-shot push all-reduce of the tensor-parallel exchange, with 2 and 4 ranks that
share cuda:0 (hipIpc inboxes between processes; handles exchanged over gloo) -- against the fp32 sum in rank order rounded once
(its contract: identical bits on every rank) and against gloo's all_reduce of the same tensors; eager and replayed from a HIP
graph; many calls in a row (epoch / parity bookkeeping)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gbx_lm_amd import tp
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ar = tp.P2PAllReduce(rank, world, 4096, dev)
        worst_gloo = 0.0
        for dt in (torch.bfloat16, torch.float16):
            for it, n in enumerate([4096, 4096, 1024, 4096, 2, 4096, 512, 4096, 4096, 4096]):
                g = torch.Generator().manual_seed(1000 * it + 17)
                xs = [(torch.randn(n, generator=g) * (1 + r)).to(dt) for r in range(world)]        # every rank knows every input
                want = sum(x.float() for x in xs[1:]) if False else None
                acc = torch.zeros(n)
                for x in xs:                                   # rank order, fp32
                    acc = acc + x.float()
                want = acc.to(dt)
                mine = xs[rank].to(dev).clone()
                ar(mine)
                assert torch.equal(mine.cpu(), want), (rank, str(dt), it, n)
                ref = xs[rank].to(dev).clone()
                dist.all_reduce(ref)                           # gloo on CUDA tensors
                worst_gloo = max(worst_gloo, float((ref.float() - mine.float()).abs().max() / (mine.float().abs().max() + 1e-6)))
                if world == 2:                                 # two addends: one rounding either way
                    assert torch.equal(ref, mine)
        # replayed from a HIP graph: the epoch is device state
        x0 = (torch.arange(4096) % 13 - 6).to(torch.bfloat16).to(dev) * (rank + 1)
        buf = x0.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ar(buf)                                            # warm-up outside capture
        torch.cuda.synchronize()
        dist.barrier()
        buf.copy_(x0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ar(buf)
        exp = (torch.arange(4096) % 13 - 6).float() * sum(r + 1 for r in range(world))
        for _ in range(5):
            buf.copy_(x0)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(buf.float().cpu(), exp.to(torch.bfloat16).float())
            dist.barrier()
        ar.close()
        out[rank] = worst_gloo
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_allreduce_shared_gpu(world):
    assert torch.cuda.is_available()
    port = 28500 + world * 10 + (os.getpid() % 1000)
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for r, v in out.items():
        assert v < 0.02, f"rank {r}: P2P vs gloo relative difference {v}"      # gloo rounds after every hop: a few bf16 ulps at 4 ranks
