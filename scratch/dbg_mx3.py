import sys, numpy as np, torch
sys.path.insert(0, '.')
DEV = 'cuda:0'
import importlib.util
spec = importlib.util.spec_from_file_location("oracle_mod", "oracle/gba_oracle.py"); oracle = importlib.util.module_from_spec(spec); spec.loader.exec_module(oracle)
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dt = torch.bfloat16
k, n = 512, 16
d = oracle.synth_layer(n, k, 2, 128, seed=3, act_dtype="bf16")
m = QuantizedLinear(k, n, False, 128, 2)
m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
sc = torch.ones_like(d["scales"]); ze = torch.zeros_like(d["zeros"])
m._buffers["scales"], m._buffers["zeros"] = sc.to(DEV), ze.to(DEV)
q = oracle.unpack(d["qweight"], 2).astype(np.float64)      # [n, k]
def run(x):
    y = torch.zeros(n, dtype=torch.float32, device=DEV).to(dt)
    ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=ops.pack_scales_zeros(m)), x.device)
    return y.float().cpu().numpy()
# 1. x = one-hot at position p (value 1.0): y[n] should be q[n, p]
for p in (0, 1, 2, 3, 15, 16, 17, 31, 32, 64, 65, 100, 127, 128, 129, 200, 300, 511):
    x = torch.zeros(k, dtype=dt, device=DEV); x[p] = 1.0
    y = run(x)
    ok = np.allclose(y, q[:, p])
    print("one-hot", p, "ok" if ok else "BAD", "got", y[:8], "want", q[:8, p])
