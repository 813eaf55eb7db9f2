"""CPU baseline leg of bench.py (child process, no GPU).  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Times, on this box's host cores, a BOUNDED sample of the bench workload and extrapolates to one decoded token:

    sample  = the seven QuantizedLinear calls (M = 1) of one decoder layer at each bit width present in the strategy
              + one lm_head matvec
    t_token = sum_bits  n_layers(bits) * t_layer(bits)  +  t_head

Three timings of the same sample:
  * value                 -- the PyTorch-CPU re-expression of the identical GBA arithmetic: oracle unpack (numpy uint32
                             bit-stream) -> w = q * scales + zeros (fp32) -> torch.matmul(x, w.T), bf16 head,
                             torch.set_num_threads(os.cpu_count())
  * cached_dequant_tok_s  -- the same with w kept dequantised (matmul only: separates unpack cost from GEMV cost)
  * c_port_tok_s          -- the oracle's C restatement (oracle/gba_oracle.c, OpenMP over output rows, fused unpack + dot)

Prints one JSON object.  Usage: python oracle/cpu_baseline.py '<json spec>'
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    spec = json.loads(sys.argv[1])
    import ctypes
    import subprocess
    subprocess.check_call(["make", "-C", HERE, "-s"])
    lib = ctypes.CDLL(os.path.join(HERE, "_build", "libgba_oracle.so"))
    i64, i32, p = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
    lib.gba_oracle_qlinear.argtypes = [p, p, p, p, p, p, i64, i64, i64, i32, i32, i32, i32, i32]
    rng = np.random.default_rng(0)
    D, I, V = spec["hidden"], spec["inter"], spec["vocab"]
    kv = spec["kv_dim"]
    shapes = [(D, D), (kv, D), (kv, D), (D, D), (I, D), (I, D), (D, I)]      # (N, K): q k v o gate up down
    g = spec["group"]
    t_layer = {}
    for bits in sorted(set(spec["layer_bits"])):
        mats = []
        for n, k in shapes:
            qw = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
            sc = np.full((n, k // g), 0x3C00, dtype=np.uint16)
            ze = np.full((n, k // g), 0xBC00, dtype=np.uint16)
            x = np.full((1, k), 0x3F80, dtype=np.uint16)
            y = np.empty((1, n), dtype=np.float64)
            mats.append((qw, sc, ze, x, y, n, k))
        best = None
        for it in range(4):                                   # first pass = warm-up
            t0 = time.perf_counter()
            for qw, sc, ze, x, y, n, k in mats:
                rc = lib.gba_oracle_qlinear(x.ctypes.data, qw.ctypes.data, sc.ctypes.data, ze.ctypes.data, None,
                                            y.ctypes.data, 1, n, k, bits, g, 1, 1, 0)
                assert rc == 0
            dt = time.perf_counter() - t0
            if it > 0:
                best = dt if best is None else min(best, dt)
        t_layer[bits] = best
    head = rng.standard_normal((V, D), dtype=np.float32)
    xh = rng.standard_normal(D, dtype=np.float32)
    head @ xh
    t0 = time.perf_counter()
    for _ in range(3):
        head @ xh
    t_head = (time.perf_counter() - t0) / 3
    t_token = sum(t_layer[b] for b in spec["layer_bits"]) + t_head
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))

    # ---- the PyTorch-CPU path (the baseline SURVEY.md 8d specifies) on the same sample ----
    import torch
    from oracle import gba_oracle as O
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    tt_full, tt_cached = {}, {}
    budget_t0 = time.perf_counter()
    for bits in sorted(set(spec["layer_bits"])):
        full = cached = 0.0
        for n, k in shapes:
            qw = rng.integers(0, 2**32, size=(n, k * bits // 32), dtype=np.uint64).astype(np.uint32)
            sc = torch.full((n, k // g), 0.01, dtype=torch.bfloat16)
            ze = torch.full((n, k // g), -0.015, dtype=torch.bfloat16)
            x = torch.randn(1, k, dtype=torch.bfloat16)
            t0 = time.perf_counter()
            q = torch.from_numpy(O.unpack(qw, bits).astype(np.float32))                   # [N, K]
            w = (q.view(n, k // g, g) * sc.float()[:, :, None] + ze.float()[:, :, None]).view(n, k)
            y = torch.matmul(x.float(), w.t())
            full += time.perf_counter() - t0
            torch.matmul(x.float(), w.t())
            t0 = time.perf_counter()
            for _ in range(3):
                y = torch.matmul(x.float(), w.t())
            cached += (time.perf_counter() - t0) / 3
            del q, w, y
        tt_full[bits], tt_cached[bits] = full, cached
    hb = torch.randn(V, D, dtype=torch.float32).to(torch.bfloat16)
    xb = torch.randn(D, dtype=torch.bfloat16)
    torch.mv(hb, xb)
    t0 = time.perf_counter()
    for _ in range(3):
        torch.mv(hb, xb)
    tt_head = (time.perf_counter() - t0) / 3
    tok_full = sum(tt_full[b] for b in spec["layer_bits"]) + tt_head
    tok_cached = sum(tt_cached[b] for b in spec["layer_bits"]) + tt_head
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        cpu_model = "unknown"
    print(json.dumps({"value": 1.0 / tok_full, "unit": "tok/s", "cores": ncores, "kind": "port",
                      "cached_dequant_tok_s": 1.0 / tok_cached, "c_port_tok_s": 1.0 / t_token, "c_port_threads": threads,
                      "cpu_model": cpu_model, "torch": torch.__version__, "sample_cpu_seconds": time.perf_counter() - budget_t0,
                      "sample": "PyTorch-CPU re-expression (oracle unpack -> fp32 dequant -> torch.matmul, bf16 torch.mv head, "
                                f"{ncores} threads) timed ONCE on the 7 QuantizedLinear calls (M = 1) of one decoder layer per bit "
                                f"width {sorted(tt_full)} (s per layer: { {b: round(v, 3) for b, v in tt_full.items()} }; matmul only with "
                                f"cached dequant, ms: { {b: round(v * 1e3, 2) for b, v in tt_cached.items()} }; head {tt_head * 1e3:.2f} ms); "
                                f"token = sum over the {len(spec['layer_bits'])} layers + head; norms/RoPE/attention excluded.  "
                                f"c_port_tok_s: oracle/gba_oracle.c (fp32, OpenMP x{threads}), ms per layer "
                                f"{ {b: round(v * 1e3, 2) for b, v in t_layer.items()} } + fp32 numpy head {t_head * 1e3:.2f} ms"}))


if __name__ == "__main__":
    main()
