import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: exercises libgbx_experimental.so (opt-in kernels outside the product ABI; "
                                       "always combined with gpu, so -m gpu still selects them)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never pass silently without a GPU: on a box with no device they fail, not skip,
    unless the run deselected them with -m 'not gpu'."""
    return


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """Every test draws its random inputs from a seed derived from its own node id: the same inputs on every run and in any
    order (a tolerance test must not pass or fail with the luck of the draw)."""
    import zlib
    import numpy as np
    import torch
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    torch.manual_seed(seed)          # seeds the CPU generator and, lazily, every CUDA device's default generator
    np.random.seed(seed % (2 ** 32))
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import gba_oracle
    return gba_oracle


@pytest.fixture(scope="session")
def c_oracle():
    from oracle import c_oracle as co
    co.build()
    return co


def oracle_matmul(x, w, scales, biases, transpose=True, group_size=64, bits=4, bias=None, out=None):
    """CPU stand-in for gbx_lm_amd.ops.quantized_matmul backed by the oracle.  TEST USE ONLY: lets the host
    graph / generate loop run on CPU so host logic is covered without a GPU."""
    import numpy as np
    import torch
    from oracle import gba_oracle as O
    qw = w.detach().cpu().view(torch.int32).numpy().view(np.uint32)
    y = O.qlinear_ref(x.detach().cpu().reshape(-1, x.shape[-1]), qw, scales.cpu(), biases.cpu(), bits, group_size,
                      bias=None if bias is None else bias.cpu())
    yt = torch.from_numpy(y.astype(np.float32)).to(x.dtype).reshape(*x.shape[:-1], w.shape[0])
    return yt


def oracle_gather_matmul(x, w, scales, biases, rhs_indices, transpose=True, group_size=64, bits=4,
                         sorted_indices=False, bias=None):
    """CPU stand-in for gbx_lm_amd.ops.gather_quantized_matmul backed by the oracle.  TEST USE ONLY."""
    import numpy as np
    import torch
    from oracle import gba_oracle as O
    qw = w.detach().cpu().view(torch.int32).numpy().view(np.uint32)
    y = O.gather_qlinear_ref(x.detach().cpu(), qw, scales.cpu(), biases.cpu(), rhs_indices.cpu().numpy(), bits, group_size,
                             bias=None if bias is None else bias.cpu())
    return torch.from_numpy(y.astype(np.float32)).to(x.dtype)


@pytest.fixture
def cpu_oracle_op(monkeypatch):
    """Route QuantizedLinear / QuantizedSwitchLinear forward through the oracle on CPU (host-logic tests only)."""
    from gbx_lm_amd import ops
    monkeypatch.setattr(ops, "quantized_matmul", oracle_matmul)
    monkeypatch.setattr(ops, "gather_quantized_matmul", oracle_gather_matmul)
    return oracle_matmul
