"""B = 32: q|k|v as three segments in one launch (today) vs one 6144-row matrix k-split into fp32 slices (no summing launch)."""
import sys, os, torch, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops
dev = 'cuda:0'
m, k, b, g = 32, 4096, 2, 128
def mk(n):
    o = types.SimpleNamespace()
    o.qweight = torch.randint(0, 2**32, (n, k // 32 * b), dtype=torch.int64, device=dev).to(torch.int32).view(torch.uint32)
    o.scales = (torch.rand((n, k // g), device=dev) * 0.01).to(torch.bfloat16); o.zeros = (-o.scales * 1.5).to(torch.bfloat16)
    o.bias = None; o.bits = b; o.group_size = g; o.input_dims = k; o.output_dims = n
    return o
copies = 40
sets = [(mk(4096), mk(1024), mk(1024), mk(6144)) for _ in range(copies)]
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
def timeit(fn, iters=40):
    for i in range(3): fn(i)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(iters): fn(i)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
t_multi = timeit(lambda i: ops.quantized_matmul_multi(x, sets[i % copies][:3]))
r = ops.quantized_matmul_partials(x, sets[0][3])
print("partials:", None if r is None else (r[0].shape, r[1]))
t_part = timeit(lambda i: ops.quantized_matmul_partials(x, sets[i % copies][3]))
print(f"q|k|v multi {t_multi:.2f} us   one matrix, k-split partials {t_part:.2f} us")
