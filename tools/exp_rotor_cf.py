#!/usr/bin/env python3
"""Rotor sweeps: overrelaxation in closed form (default) against sweep by sweep (MLMCPI_OR_KERNEL=block): largest angle
difference of whole draws, and the time of a 10 + 1 draw with the QoI at M = 65536 x 1024 chains."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7
def angle_diff(a, b):
    d = (a - b).abs()
    return float(torch.minimum(d, (d - 2 * np.pi).abs()).max())
def draw(act, x0, n_or, n_hb, kern):
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    try:
        x = x0.clone()
        ops.path_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 5)
        return x
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")
for M, B in ((64, 3), (1024, 3), (4096, 2), (65536, 2), (6, 2)):
    act = abi.path_action(abi.ROTOR, M, M / 16.0, 0.25)
    x0 = ops.path_initialise(act, B, SEED)
    for n_or, n_hb in ((1, 0), (2, 0), (7, 0), (10, 0), (16, 0), (17, 0), (35, 0), (1, 1), (10, 1), (10, 2), (20, 1)):
        a = draw(act, x0, n_or, n_hb, "block")
        p = draw(act, x0, n_or, n_hb, "")
        print(f"M={M} B={B} ({n_or},{n_hb}): max |closed - block| = {angle_diff(a, p):.3e}", flush=True)
def run(kern, steps=20):
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    act = abi.path_action(abi.ROTOR, 65536, 4096.0, 0.25)
    x = ops.path_initialise(act, 1024, SEED)
    w = torch.empty_like(x)
    s = 0
    for _ in range(3):
        x, w, q = ops.path_sweep_draw_qoi(act, x, w, x, 10, 1, SEED, 0, s)
        s += 11
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        x, w, q = ops.path_sweep_draw_qoi(act, x, w, x, 10, 1, SEED, 0, s)
        s += 11
    torch.cuda.synchronize(); abi.set_option("MLMCPI_OR_KERNEL", "")
    return (time.perf_counter() - t0) / steps * 1e3, float(q.mean())
for rep in range(2):
    tb, qb = run("block"); tc, qc = run("")
    print(f"10 + 1 + QoI: block {tb:.4f} ms  closed {tc:.4f} ms  ratio {tc / tb:.3f}  {65536 * 1024 * 11 / tc / 1e6:.1f} G/s   q {qb:.5f} {qc:.5f}", flush=True)
