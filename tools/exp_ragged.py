#!/usr/bin/env python3
"""Lattices no tile divides: time of the overrelaxation part and of the heat-bath part of a 10 + 1 draw, closed form (masked
edge tiles) against the sweep-by-sweep kernels (MLMCPI_OR_KERNEL=block)."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 5


def timed(act, B, n_or, n_hb, kern, qoi, reps=10):
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    try:
        x = ops.lattice_initialise(act, B, SEED, 0)
        w = torch.empty_like(x)
        s = 0
        def step():
            nonlocal x, w, s
            if qoi:
                x, w, _ = ops.lattice_sweep_draw_qoi(act, x, w, x, n_or, n_hb, SEED, 0, s, 1)
            else:
                ops.lattice_sweep_draw(act, x, w, n_or, n_hb, SEED, 0, s)
            s += n_or + n_hb
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    finally:
        abi.set_option("MLMCPI_OR_KERNEL", "")


for Mt, Mx, B in ((130, 70, 4096), (1000, 1000, 32), (200, 136, 2048), (96, 96, 4096)):
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    for kern in ("", "block"):
        t_or = timed(act, B, 10, 0, kern, False)
        t_hb = timed(act, B, 0, 1, kern, False)
        t_all = timed(act, B, 10, 1, kern, True)
        rate = 2 * Mt * Mx * 11 * B / (t_all * 1e-3) / 1e9
        print(f"{Mt}x{Mx} B={B} kernel={kern or 'perm'}: 10 OR {t_or:.3f} ms, 1 HB {t_hb:.3f} ms, 10+1+QoI {t_all:.3f} ms = {rate:.0f} G/s", flush=True)
