#!/usr/bin/env python3
"""A/B of the GFF draw's last launches: gff_or_heat_kernel<5> against two launches (MLMCPI_OR_HEAT=split), 512^2 x 1024 chains."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7
for M, B, steps in ((512, 1024, 10), (512, 64, 40), (512, 4, 100)):
    out = {}
    for mode in ("split", "fused"):
        abi.set_option("MLMCPI_OR_HEAT", mode)
        act = abi.lattice_action(abi.GFF, M, M, mass=10.0)
        x = ops.lattice_initialise(act, B, SEED, 0)
        w = torch.empty_like(x)
        s = 0
        for _ in range(3):
            x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, SEED, 0, s, 3)
            s += 11
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, SEED, 0, s, 3)
            s += 11
        torch.cuda.synchronize()
        out[mode] = ((time.perf_counter() - t0) / steps * 1e3, x, float(q.mean()))
    print(f"M={M} B={B}: split {out['split'][0]:.4f} ms fused {out['fused'][0]:.4f} ms ratio {out['fused'][0] / out['split'][0]:.3f} "
          f"G/s fused {M * M * 11 * B / out['fused'][0] / 1e6:.1f} identical {torch.equal(out['split'][1], out['fused'][1])} q {out['fused'][2]:.5f}", flush=True)
