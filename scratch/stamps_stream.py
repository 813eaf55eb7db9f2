"""In-kernel s_memrealtime timeline of the streamed GEMV (diagnostic build -DGBX_STAMPS, never the shipped .so):
   scratch/build_exp.sh stamps gemv_stream.hip -DGBX_STAMPS
   GBX_QLINEAR_LIB=scratch/exp/lib_stamps.so python scratch/stamps_stream.py [first_op] [nops]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
stamps = torch.zeros(256 * 8 * 64, dtype=torch.int64, device=dev)
os.environ["GBX_STAMP_PTR"] = str(stamps.data_ptr())
from gbx_lm_amd import ops, synthetic as S  # noqa: E402
first, nops = (int(sys.argv[1]) if len(sys.argv) > 1 else 2), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", cfg["num_hidden_layers"], 128), device=dev, seed=0)
dt = torch.bfloat16
D, I = cfg["hidden_size"], cfg["intermediate_size"]
a0 = model.model.layers[0].self_attn
mk = lambda n: torch.randn(n, device=dev).to(dt)
hb, qb, kb, vb, ab, actb, out_h = mk(D), mk(D), mk(1024), mk(1024), mk(D), mk(I), mk(D)
descs, keep = [], []
for blk in model.model.layers:
    at, mlp = blk.self_attn, blk.mlp
    pk = lambda *mods: torch.cat([ops.pack_scales_zeros(m) for m in mods])
    p_qkv, p_o, p_dn = pk(at.q_proj, at.k_proj, at.v_proj), pk(at.o_proj), pk(mlp.down_proj)
    p_gu = ops.pack_scales_zeros(mlp.gate_proj, mlp.up_proj)
    keep.append((p_qkv, p_o, p_gu, p_dn))
    descs += [
        ops.make_fused_desc(hb, (at.q_proj, at.k_proj, at.v_proj), (qb, kb, vb), norm_weight=blk.input_layernorm.weight, eps=1e-5, packed=p_qkv),
        ops.make_fused_desc(ab, (at.o_proj,), (out_h,), residual=hb, packed=p_o),
        ops.make_fused_desc(hb, (mlp.gate_proj, mlp.up_proj), (actb,), norm_weight=blk.post_attention_layernorm.weight, eps=1e-5, swiglu=True, packed=p_gu),
        ops.make_fused_desc(actb, (mlp.down_proj,), (out_h,), residual=hb, packed=p_dn),
    ]
prog = ops.StreamProgram(descs, dev, dt)
for L in range(3, 12):            # 2-bit layers; the last one's stamps are read
    stamps.zero_()
    prog.run(4 * L + first, 4 * L + first + nops)
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 64).astype(np.float64)
t0 = s[:, 63].min()
rel = (s - t0) / 100.0
print(f"ops [{first}, {first + nops}) of a 2-bit layer: kernel span {rel[:, 62].max():.2f} us; wave start spread {rel[:, 63].max():.2f} us")
names = ["op entry", "x+ring issued", "norm reduced", "x image (barrier)", "lane x ready", "1st slot landed", "1st slot computed",
         "1st refill issued", "loop done", "barrier", "epilogue done"]
for o in range(nops):
    print(f" op {o}:")
    for i, nm in enumerate(names):
        c = rel[:, o * 12 + i]
        print(f"  {nm:20s} min {c.min():6.2f}  med {np.median(c):6.2f}  max {c.max():6.2f}")
c = rel[:, 62]
print(f"  {'kernel end':20s} min {c.min():6.2f}  med {np.median(c):6.2f}  max {c.max():6.2f}")
