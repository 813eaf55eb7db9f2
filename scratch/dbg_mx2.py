import sys, numpy as np, torch
sys.path.insert(0, '.')
DEV = 'cuda:0'
import importlib.util
spec = importlib.util.spec_from_file_location("oracle_mod", "oracle/gba_oracle.py"); oracle = importlib.util.module_from_spec(spec); spec.loader.exec_module(oracle)
from gbx_lm_amd import ops
from gbx_lm_amd.models.quantized_linear_gba import QuantizedLinear
dt = torch.bfloat16
def mod(n, k, seed):
    d = oracle.synth_layer(n, k, 2, 128, seed=seed, act_dtype="bf16")
    m = QuantizedLinear(k, n, False, 128, 2)
    m._buffers["qweight"] = torch.from_numpy(d["qweight"].view(np.int32)).view(torch.uint32).to(DEV)
    m._buffers["scales"], m._buffers["zeros"] = d["scales"].to(DEV), d["zeros"].to(DEV)
    return m, d
for k in (512, 1024, 2048, 4096, 8192, 14336):
    for n in (16, 4096):
        m, d = mod(n, k, 3)
        torch.manual_seed(0)
        x = torch.randn(k).to(dt).to(DEV)
        y = torch.zeros(n, dtype=dt, device=DEV)
        ops.qlinear_fused(ops.make_fused_desc(x, (m,), (y,), packed=ops.pack_scales_zeros(m)), x.device)
        y64 = oracle.qlinear_ref(x.cpu()[None], d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)[0]
        err = np.abs(y.cpu().double().numpy() - y64).max() / np.abs(y64).max()
        # per-group probe: x nonzero only in group g
        bad = []
        if n == 16:
            for g in range(k // 128):
                xg = torch.zeros(k, dtype=dt, device=DEV); xg[g * 128:(g + 1) * 128] = x[g * 128:(g + 1) * 128]
                yg = torch.zeros(n, dtype=dt, device=DEV)
                ops.qlinear_fused(ops.make_fused_desc(xg, (m,), (yg,), packed=ops.pack_scales_zeros(m)), x.device)
                r64 = oracle.qlinear_ref(xg.cpu()[None], d["qweight"], d["scales"], d["zeros"], 2, 128, acc=np.float64)[0]
                e = np.abs(yg.cpu().double().numpy() - r64).max() / (np.abs(r64).max() + 1e-9)
                if e > 0.02: bad.append((g, round(float(e), 2)))
        print("K", k, "N", n, "rel err %.4f" % err, "bad groups:", bad[:40])
