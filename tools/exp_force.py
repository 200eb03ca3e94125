#!/usr/bin/env python3
"""Launch time of mlmcpi_lattice_force (Schwinger 1024^2 x 32 chains; floor = state read + force written = 1.07 GB).
   MLMCPI_LIB_VARIANT=<name> python tools/exp_force.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
x = ops.lattice_initialise(act, 32, 7)
for _ in range(5):
    ops.lattice_force(act, x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.lattice_force(act, x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print("variant %-8s force %.4f ms  %.0f GB/s  %.3f of 8 TB/s" % (os.environ.get("MLMCPI_LIB_VARIANT", "") or "main", ms, 2 * 8 * x.numel() / ms / 1e6, 2 * 8 * x.numel() / ms / 1e6 / 8000))
