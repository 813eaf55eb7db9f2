import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
DEV = 'cuda:0'
import importlib.util
spec = importlib.util.spec_from_file_location("oracle_mod", "oracle/gba_oracle.py"); oracle = importlib.util.module_from_spec(spec); spec.loader.exec_module(oracle)
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
def run(bits, g, dt, n=1030, k=4096):
    act = "bf16" if dt == torch.bfloat16 else "f16"
    def mod(n, k, seed):
        d = oracle.synth_layer(n, k, bits, g, seed=seed, act_dtype=act)
        m = QuantizedLinear(k, n, False, g, bits)
        m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
        m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
        return m
    gate, up = mod(n, k, 4), mod(n, k, 5)
    pgu = ops.pack_scales_zeros(gate, up)
    torch.manual_seed(0)
    x = (torch.randn(k) * 2).to(dt).to(DEV)
    nw = (1 + 0.1 * torch.randn(k)).to(dt).to(DEV)
    def run_gu(packed):
        o = torch.zeros(n, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (gate, up), (o,), norm_weight=nw, eps=1e-5, swiglu=True, packed=packed), x.device)
        return o
    a = run_gu(None)
    outs = [run_gu(pgu) for _ in range(5)]
    for i, b in enumerate(outs):
        bad = (a != b).nonzero().flatten()
        print(bits, g, act, n, "run", i, "mismatches", len(bad), bad[:8].tolist(), (a.float() - b.float()).abs().max().item(), "self-consistent", torch.equal(b, outs[0]))
for dt in (torch.float16, torch.bfloat16):
    for bits, g in ((4, 128), (3, 64), (2, 128)):
        run(bits, g, dt)
        run(bits, g, dt, n=14336)
