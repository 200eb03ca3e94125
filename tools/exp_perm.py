#!/usr/bin/env python3
"""Closed-form overrelaxation (schwinger_perm_kernel / schwinger_perm_heat_kernel, the default) against the sweep-by-sweep
register-block kernels (MLMCPI_OR_KERNEL=block): largest angle difference of whole draws, and the time of a 10 + 1 draw with
the QoI at 1024 x 1024 x 32."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7


def angle_diff(a, b):
    d = (a - b).abs()
    d = torch.minimum(d, (d - 2 * np.pi).abs())
    return float(d.max())


def draw(act, x0, n_or, n_hb, kern, fuse=0):
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    try:
        x = x0.clone()
        ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 5, fuse=fuse)
        return x
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")


for Mt, Mx, B in ((128, 128, 2), (192, 128, 2), (64, 64, 2), (1024, 1024, 2)):
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    x0 = ops.lattice_initialise(act, B, SEED, 0)
    for n_or, n_hb in ((1, 0), (2, 0), (5, 0), (7, 0), (8, 0), (9, 0), (10, 0), (13, 0), (1, 1), (5, 1), (7, 1), (8, 1), (10, 1), (12, 1), (10, 2)):
        a = draw(act, x0, n_or, n_hb, "block")
        p = draw(act, x0, n_or, n_hb, "")
        print(f"{Mt}x{Mx} B={B} ({n_or},{n_hb}): max |perm - block| = {angle_diff(a, p):.3e}", flush=True)


def run(B, kern, steps, n_or=10):
    abi.set_option("MLMCPI_OR_KERNEL", kern)
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
    abi.set_option("MLMCPI_OR_KERNEL", "")
    return (time.perf_counter() - t0) / steps * 1e3, float(q.mean())


for B, steps in ((32, 30), (1, 300), (128, 8)):
    for rep in range(2):
        tb, qb = run(B, "block", steps)
        tp, qp = run(B, "", steps)
        print(f"B={B:4d} block {tb:.4f} ms  perm {tp:.4f} ms  ratio {tp / tb:.3f}  G/s perm {2 * 1024 * 1024 * 11 * B / tp / 1e6:.1f}  q {qb:.6f} {qp:.6f}", flush=True)
