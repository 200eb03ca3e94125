#!/usr/bin/env python3
"""Minimal counter-collection target: a few 10 + 1 draws with the QoI, Schwinger 1024 x 1024 x 32 (no child processes)."""
import sys
import torch
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "."))
from mlmcpathintegral_amd import abi, ops
abi.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
x = ops.lattice_initialise(act, B, 7, 0)
w = torch.empty_like(x)
s = 0
for _ in range(4):
    x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, 7, 0, s, 1)
    s += 11
torch.cuda.synchronize()
