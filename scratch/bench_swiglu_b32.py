import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", 32, 128), device=dev, seed=0)
layers = [b for b in model.model.layers if b.mlp.gate_proj.bits == 2]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(M, 4096, device=dev).to(torch.bfloat16)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / len(layers)
def unfused():
    for b in layers:
        g, u = ops.quantized_matmul_multi(x, (b.mlp.gate_proj, b.mlp.up_proj)); ops.swiglu(g, u)
def fused():
    for b in layers:
        a = ops.quantized_swiglu(x, b.mlp.gate_proj, b.mlp.up_proj); assert a is not None
print(f"M={M}: gate|up + swiglu unfused {timeit(unfused):.2f} us, fused {timeit(fused):.2f} us")
a = ops.quantized_swiglu(x, layers[0].mlp.gate_proj, layers[0].mlp.up_proj)
g, u = ops.quantized_matmul_multi(x, (layers[0].mlp.gate_proj, layers[0].mlp.up_proj)); w = ops.swiglu(g, u)
d = (a.float() - w.float()).abs(); print("max diff", d.max().item(), "frac differing", (d > 0).float().mean().item(), "rows differing", (d > 0).any(1).nonzero().flatten().tolist()[:40])
