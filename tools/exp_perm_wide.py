#!/usr/bin/env python3
"""Narrow (2 x 512 threads per CU, two planes at K = 10) against wide (1024 threads, one plane) workgroups of
schwinger_perm_heat_kernel at 1024 x 1024 x B."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7
def run(B, mode, steps, n_or=10):
    abi.set_option("MLMCPI_OR_HEAT", mode)
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
    abi.set_option("MLMCPI_OR_HEAT", "")
    return (time.perf_counter() - t0) / steps * 1e3, x
for n_or in (10, 5):
    for B, steps in ((32, 30), (1, 300)):
        tn, xn = run(B, "narrow", steps, n_or)
        tw, xw = run(B, "wide", steps, n_or)
        print(f"n_or={n_or} B={B:3d} narrow {tn:.4f} ms  wide {tw:.4f} ms  identical {torch.equal(xn, xw)}", flush=True)
