"""B = 32: q|k|v as ONE k-split partial-slices GEMM (N = 6144) vs the multi-segment batch GEMM it would replace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dev = torch.device("cuda:0")
def mk(n, k, seed):
    m = QuantizedLinear(k, n, bias=False, group_size=128, bits=2)
    g = torch.Generator(device=dev).manual_seed(seed)
    m._set("qweight", torch.randint(0, 2**32, (n, k // 16), dtype=torch.int64, device=dev, generator=g).to(torch.int32).view(torch.uint32))
    sc = (torch.rand((n, k // 128), device=dev, generator=g) * 0.01).to(torch.bfloat16)
    m._set("scales", sc); m._set("zeros", (-sc * 1.5).to(torch.bfloat16))
    return m
def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
x = torch.randn(32, 4096, device=dev).to(torch.bfloat16)
sets = [(mk(4096, 4096, 3 * i), mk(1024, 4096, 3 * i + 1), mk(1024, 4096, 3 * i + 2), mk(6144, 4096, 100 + i)) for i in range(40)]
i = [0]
def multi():
    q, k, v, _ = sets[i[0] % 40]; i[0] += 1
    ops.quantized_matmul_multi(x, (q, k, v))
def part():
    *_, c = sets[i[0] % 40]; i[0] += 1
    r = ops.quantized_matmul_partials(x, c)
    assert r is not None
print("multi-segment q|k|v:", round(t(multi), 2), "us;  one N=6144 partial-slices GEMM:", round(t(part), 2), "us; slices", ops.quantized_matmul_partials(x, sets[0][3])[1])
