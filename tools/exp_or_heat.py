#!/usr/bin/env python3
"""A/B of the draw's last launches: 5 overrelaxation sweeps + heat bath + QoI as one launch (schwinger_or_heat_kernel<5>)
against two (MLMCPI_OR_HEAT=split), Schwinger 1024 x 1024, whole draws of 10 OR + 1 HB + QoI through one ABI call."""
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
    dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, float(q.mean()), x


for B, steps in ((32, 30), (1, 300), (4, 100), (128, 8)):
    for rep in range(2):
        out = {}
        for mode in ("split", "fused"):
            ms, q, x = run(B, mode, steps)
            out[mode] = (ms, q, x)
        same = torch.equal(out["split"][2], out["fused"][2])
        print(f"B={B:4d} split {out['split'][0]:.4f} ms  fused {out['fused'][0]:.4f} ms  ratio {out['fused'][0] / out['split'][0]:.3f}  "
              f"G/s fused {2 * 1024 * 1024 * 11 * B / out['fused'][0] / 1e6:.1f}  identical {same}  q {out['fused'][1]:.6f}", flush=True)
