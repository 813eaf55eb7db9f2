import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S
from gbx_lm_amd.decode import DecodeRunner
from bench import make_caches
calls = {"fused": 0, "none": 0}
orig = ops.quantized_swiglu
def spy(x, g, u):
    r = orig(x, g, u); calls["fused" if r is not None else "none"] += 1
    if r is None: print("None for", x.shape, x.dtype, g.bits, g.group_size, g.scales.dtype, g.bias is None, x.is_contiguous())
    return r
ops.quantized_swiglu = spy
dev = torch.device("cuda:0")
cfg = dict(S.LLAMA3_8B, num_hidden_layers=2)
model = S.build_model(cfg, S.make_strategy("S22", 2, 128), device=dev, seed=0)
c = make_caches(model, 32, 256, dev)
r = DecodeRunner(model, c, use_graph=False)
r.step(torch.zeros(32, dtype=torch.long, device=dev))
print(calls)
