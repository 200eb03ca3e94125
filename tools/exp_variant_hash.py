#!/usr/bin/env python3
"""Bit-for-bit comparison of two library builds (MLMCPI_LIB_VARIANT): SHA-256 of the states a set of Schwinger draws leaves
behind -- closed-form launches of every depth, 64 x 64 and 64 x 32 tiles, with and without the heat bath and the QoI.
   MLMCPI_LIB_VARIANT=r04 python tools/exp_variant_hash.py > a.txt; python tools/exp_variant_hash.py > b.txt; diff a.txt b.txt"""
import hashlib, sys
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 11


def h(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]


for Mt, Mx, B, beta in ((128, 128, 3, 1.0), (192, 128, 2, 1.0), (256, 192, 2, 0.7), (192, 96, 2, 1.0), (64, 64, 2, 1.0), (128, 128, 2, 3.0), (1024, 1024, 2, 1.0)):
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=beta)
    x0 = ops.lattice_initialise(act, B, SEED, 0)
    for n_or, n_hb in ((1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (9, 0), (10, 0), (13, 0), (23, 0), (1, 1), (4, 1), (6, 1), (7, 1), (9, 1), (10, 1), (10, 2)):
        x = x0.clone()
        ops.lattice_sweep_draw(act, x, torch.empty_like(x), n_or, n_hb, SEED, 0, 5)
        line = f"{Mt}x{Mx} B={B} beta={beta} ({n_or},{n_hb}): {h(x)}"
        if n_hb:
            a, w, q = ops.lattice_sweep_draw_qoi(act, x0.clone(), torch.empty_like(x0), torch.empty_like(x0), n_or, n_hb, SEED, 0, 5, 1)
            line += f" qoi {h(a)} {h(q)}"
        print(line, flush=True)
