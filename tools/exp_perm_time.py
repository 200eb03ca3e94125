#!/usr/bin/env python3
"""Time of a 10 + 1 draw with the QoI, Schwinger 1024 x 1024, for the library named by MLMCPI_LIB_VARIANT (same-box A/B)."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7
def run(B, steps, n_or=10):
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    x = ops.lattice_initialise(act, B, SEED, 0)
    w = torch.empty_like(x)
    s = 0
    for _ in range(5):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, n_or, 1, SEED, 0, s, 1)
        s += n_or + 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, n_or, 1, SEED, 0, s, 1)
        s += n_or + 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(q.mean())
v = os.environ.get("MLMCPI_LIB_VARIANT", "") or "main"
for B, steps in ((32, 40), (128, 10), (1, 300)):
    for rep in range(2):
        t, q = run(B, steps)
        print(f"variant {v:8s} B={B:4d} {t:.4f} ms  {2 * 1024 * 1024 * 11 * B / t / 1e6:.1f} G/s  q {q:.6f}", flush=True)
