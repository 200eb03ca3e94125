#!/usr/bin/env python3
"""Launch time of ONE overrelaxation sweep (Schwinger 1024^2 x 32; floor = state read + written = 1.07 GB), register-block
kernel (MLMCPI_OR_KERNEL=block) and closed-form launch:   MLMCPI_LIB_VARIANT=<name> python tools/exp_or1_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
abi.load()
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
x = ops.lattice_initialise(act, 32, 7); w = torch.empty_like(x)
for kernel in ("block", "perm"):
    abi.set_option("MLMCPI_OR_KERNEL", kernel)
    s = 0
    for _ in range(5):
        x, w = ops.lattice_sweep_draw_pingpong(act, x, w, 1, 0, 7, 0, s, 1); s += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        x, w = ops.lattice_sweep_draw_pingpong(act, x, w, 1, 0, 7, 0, s, 1); s += 1
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print("variant %-6s %-5s %.4f ms  %.3f of 8 TB/s" % (os.environ.get("MLMCPI_LIB_VARIANT", "") or "main", kernel, ms, 2 * 8 * x.numel() / ms / 1e6 / 8000), flush=True)
