"""CPU baseline leg of bench.py (child process, no torch, no GPU).  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Times the oracle's C restatement (oracle/gba_oracle.c, OpenMP over output rows) of the hot path on this box's host
cores for a BOUNDED sample of the bench workload and extrapolates to one decoded token:

    sample  = the seven QuantizedLinear calls (M = 1) of one decoder layer at each bit width present in the strategy
              + one bf16->fp32 lm_head matvec (numpy)
    t_token = sum_bits  n_layers(bits) * t_layer(bits)  +  t_head

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
    print(json.dumps({"value": 1.0 / t_token, "unit": "tok/s", "cores": threads, "kind": "port",
                      "sample": "oracle/gba_oracle.c (fp32, OpenMP) timed on the 7 QuantizedLinear calls of one decoder "
                                f"layer per bit width {sorted(t_layer)} (min of 3, ms: "
                                f"{ {b: round(v * 1e3, 2) for b, v in t_layer.items()} }) + fp32 lm_head matvec "
                                f"({t_head * 1e3:.2f} ms, numpy); token = sum over the {len(spec['layer_bits'])} layers "
                                "+ head; norms/RoPE/attention excluded"}))


if __name__ == "__main__":
    main()
