#!/usr/bin/env python3
"""Shape fuzz of the closed-form Schwinger overrelaxation kernels against the sweep-by-sweep ones (MLMCPI_OR_KERNEL=block; for
lattices that 64 x 64 tiles do not divide the 2 x 2 patch / generic kernels run there): every lattice shape the closed form
accepts up to 320 x 256, every depth 1 .. 13, with and without a heat-bath sweep behind; and the rotor's 1-D form."""
import itertools, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 11
def adiff(a, b):
    d = (a - b).abs()
    return float(torch.minimum(d, (d - 2 * np.pi).abs()).max())
worst = 0.0
n = 0
# r05: + lattices no tile divides (masked edge tiles), the smallest extents the closed form takes, extents around the
# windows of the fused launch (66 ... 70, 126 ... 134)
RAGGED = [(64, 34), (66, 32), (66, 34), (70, 70), (68, 128), (126, 130), (130, 126), (134, 66), (200, 72), (250, 250), (322, 130), (64, 250)]
for Mt, Mx in list(itertools.product((64, 128, 192, 320), (32, 64, 96, 128, 160, 256))) + RAGGED:
    for beta in (1.0, 3.0, 5.0):   # step envelope (2 beta <= 8), and the wrapped-Cauchy sampler beyond
        act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=beta)
        B = 1 + (Mt // 64 + Mx // 32) % 3
        x0 = ops.lattice_initialise(act, B, SEED, 0)
        for n_or, n_hb in ((1, 0), (3, 1), (7, 0), (8, 1), (10, 1), (10, 0), (13, 1)):
            res = {}
            for kern in ("block", ""):
                abi.set_option("MLMCPI_OR_KERNEL", kern)
                try:
                    x = x0.clone()
                    ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 3)
                    res[kern] = x
                finally:
                    abi.set_option("MLMCPI_OR_KERNEL", "")
            d = adiff(res["block"], res[""])
            tol = 5e-13 if n_hb == 0 else 2e-10
            n += 1
            worst = max(worst, d if n_hb == 0 else 0.0)
            if d > tol:
                print(f"MISMATCH {Mt}x{Mx} beta={beta} B={B} ({n_or},{n_hb}): {d:.3e}", flush=True)
print(f"schwinger: {n} cases, largest overrelaxation-only difference {worst:.3e}", flush=True)
worst, n = 0.0, 0
for M in (2, 4, 6, 64, 1000, 2048, 4100, 65536):
    act = abi.path_action(abi.ROTOR, M, M / 8.0, 0.25)
    x0 = ops.path_initialise(act, 2, SEED)
    for n_or, n_hb in ((1, 0), (2, 1), (9, 0), (16, 0), (17, 1), (40, 0)):
        res = {}
        for kern in ("block", ""):
            abi.set_option("MLMCPI_OR_KERNEL", kern)
            try:
                x = x0.clone()
                ops.path_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 3)
                res[kern] = x
            finally:
                abi.set_option("MLMCPI_OR_KERNEL", "")
        d = adiff(res["block"], res[""])
        n += 1
        worst = max(worst, d if n_hb == 0 else 0.0)
        if d > (5e-13 if n_hb == 0 else 5e-9):
            print(f"MISMATCH rotor M={M} ({n_or},{n_hb}): {d:.3e}", flush=True)
print(f"rotor: {n} cases, largest overrelaxation-only difference {worst:.3e}", flush=True)

# GFF register-block kernels on 32 x 32 tiles with masked edges (r05) against the LDS-resident sweep-by-sweep kernels: bit for bit
worst, n = 0, 0
for M in (64, 66, 70, 96, 100, 130, 190, 250):
    act = abi.lattice_action(abi.GFF, M, M, mass=3.0)
    x0 = ops.lattice_initialise(act, 2, SEED, 0)
    for n_or, n_hb in ((1, 0), (5, 0), (6, 1), (10, 1), (3, 2)):
        res = {}
        for kern in ("lds", ""):
            abi.set_option("MLMCPI_OR_KERNEL", kern)
            try:
                x = x0.clone()
                ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 3)
                res[kern] = x
            finally:
                abi.set_option("MLMCPI_OR_KERNEL", "")
        n += 1
        if not torch.equal(res["lds"], res[""]):
            worst += 1
            print(f"MISMATCH gff {M}x{M} ({n_or},{n_hb}): {float((res['lds'] - res['']).abs().max()):.3e}", flush=True)
print(f"gff: {n} cases, {worst} not bit-identical", flush=True)
