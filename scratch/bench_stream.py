"""Per-op timing: launch-per-op fused GEMV (gbx_qlinear_fused_forward) vs the streamed form (gbx_stream_run), Llama-3-8B
shapes, synthetic S22 weights, every 2-bit layer's own weights cycled (>> Infinity Cache).  Usage: python scratch/bench_stream.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gbx_lm_amd import ops, synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
strategy = sys.argv[1] if len(sys.argv) > 1 else "S22"
cfg = S.LLAMA3_8B
model = S.build_model(cfg, S.make_strategy(strategy, cfg["num_hidden_layers"], 128), device=dev, seed=0)
dt = torch.bfloat16
D, I = cfg["hidden_size"], cfg["intermediate_size"]
a0 = model.model.layers[0].self_attn
mk = lambda n: torch.randn(n, device=dev).to(dt)
hb, qb, kb, vb, ab, actb = mk(D), mk(a0.n_heads * a0.head_dim), mk(a0.n_kv_heads * a0.head_dim), mk(a0.n_kv_heads * a0.head_dim), mk(D), mk(I)
out_h = mk(D)
descs, keep, bits_of = [], [], []
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
    bits_of.append(mlp.gate_proj.bits)
prog = ops.StreamProgram(descs, dev, dt)
names = ["norm+qkv", "o+res", "norm+gate/up+swiglu", "down+res"]


def timeit(fn, n_launch, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_launch)


for bits in sorted(set(bits_of)):
    layers = [i for i, b in enumerate(bits_of) if b == bits]
    print(f"--- {bits}-bit layers ({len(layers)}) ---")
    tot_old = tot_new = 0.0
    for t in range(4):
        def old():
            for L in layers:
                ops.qlinear_fused(descs[4 * L + t], dev)

        def new():
            for L in layers:
                prog.run(4 * L + t, 4 * L + t + 1)
        a, b = timeit(old, len(layers)), timeit(new, len(layers))
        tot_old += a
        tot_new += b
        print(f"{names[t]:24s} launch {a:7.2f} us   stream {b:7.2f} us")

    def pair():
        for L in layers:
            prog.run(4 * L + 2, 4 * L + 4)

    def quad():
        for L in layers:
            prog.run(4 * L, 4 * L + 4)
    print(f"{'sum of four':24s} launch {tot_old:7.2f} us   stream {tot_new:7.2f} us")
    print(f"gate/up + down in ONE launch (timing only, no edge): {timeit(pair, len(layers)):7.2f} us")
    print(f"all four in ONE launch (timing only, no edge):       {timeit(quad, len(layers)):7.2f} us")
