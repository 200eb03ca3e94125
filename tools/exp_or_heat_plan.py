#!/usr/bin/env python3
"""Which split of 10 overrelaxation sweeps around the fused last launch: (a | b + HB) as two ABI calls, Schwinger 1024^2."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7


def run(B, first, last, steps):
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    x = ops.lattice_initialise(act, B, SEED, 0)
    w = torch.empty_like(x)
    s = 0

    def draw():
        nonlocal x, w, s
        cur, oth = x, w
        for d in first:
            cur, oth = ops.lattice_sweep_draw_pingpong(act, cur, oth, d, 0, SEED, 0, s, d)
            s += d
        x, w, q = ops.lattice_sweep_draw_qoi(act, cur, oth, cur, last, 1, SEED, 0, s, 1, max(last, 1))
        s += last + 1
        return q
    for _ in range(5):
        draw()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        q = draw()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(q.mean())


for B, steps in ((32, 30), (1, 300)):
    for first, last in (((5,), 5), ((6,), 4), ((5, 5), 0), ((4, 3), 3), ((5, 4), 1), ((5, 3), 2), ((5, 2), 3)):
        ms, q = run(B, first, last, steps)
        print(f"B={B:3d} plan {first} | {last}+HB: {ms:.4f} ms  {2 * 1024 * 1024 * 11 * B / ms / 1e6:.1f} G/s  q {q:.6f}", flush=True)
