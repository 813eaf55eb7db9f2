"""Does the Infinity Cache make a weight-streaming GEMV faster when another kernel has just read the same weights?
For every 2-bit layer: touch(L) = a reduction over gate|up qweight (reads every byte, default cache policy), then the fused
gate/up launch of L.  gate/up time warm = (touch + gate/up loop) - (touch-only loop); cold = the launch alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S
dev = torch.device("cuda:0")
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy("S22", 32, 128), device=dev, seed=0)
dt = torch.bfloat16
hb = torch.randn(4096, device=dev).to(dt); actb = torch.empty(14336, device=dev, dtype=dt); outh = torch.empty(4096, device=dev, dtype=dt)
layers = [b for b in model.model.layers if b.mlp.gate_proj.bits == 2]
keep, d_gu, d_dn = [], [], []
for blk in layers:
    mlp = blk.mlp
    p = ops.pack_scales_zeros(mlp.gate_proj, mlp.up_proj); pd = ops.pack_scales_zeros(mlp.down_proj); keep += [p, pd]
    d_gu.append(ops.make_fused_desc(hb, (mlp.gate_proj, mlp.up_proj), (actb,), norm_weight=blk.post_attention_layernorm.weight, eps=1e-5, swiglu=True, packed=p))
    d_dn.append(ops.make_fused_desc(actb, (mlp.down_proj,), (outh,), residual=hb, packed=pd))
sink = torch.zeros(64, device=dev, dtype=torch.int64)
def touch(mods):
    for m in mods:
        sink[0] += m.qweight.view(torch.int32).sum()
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
for name, descs, mods_of in (("gate/up", d_gu, lambda b: (b.mlp.gate_proj, b.mlp.up_proj)), ("down", d_dn, lambda b: (b.mlp.down_proj,))):
    cold = timeit(lambda: [ops.qlinear_fused(d, dev) for d in descs])
    t_only = timeit(lambda: [touch(mods_of(b)) for b in layers])
    both = timeit(lambda: [(touch(mods_of(b)), ops.qlinear_fused(d, dev)) for b, d in zip(layers, descs)])
    print(f"{name}: cold {cold:.2f} us; touch-only {t_only:.2f} us; touch+launch {both:.2f} us -> warm launch ~{both - t_only:.2f} us")
