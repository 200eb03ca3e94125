#!/usr/bin/env python3
"""Phase timeline of schwinger_perm_heat_kernel from in-kernel wall-clock stamps (instrumentation build:
tools/build_variant.sh WORK stamps -DMLMCPI_STAMPS; MLMCPI_LIB_VARIANT=stamps).  Schwinger 1024^2, draws of 10 OR + 1 HB + QoI."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
lib = abi.load()
SEED = 7
ORDER = [0, 1, 2, 10, 3, 4, 5, 6, 7, 8, 9]
NAMES = ["theta_0 loads + plane A", "gather A", "plane B", "gather B", "image down", "HB mu=0 even", "HB mu=0 odd", "HB mu=1 even", "HB mu=1 odd", "write-out + QoI"]
n_or = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for B in (32, 1):
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    x = ops.lattice_initialise(act, B, SEED, 0)
    w = torch.empty_like(x)
    s = 0
    for _ in range(4):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, n_or, 1, SEED, 0, s, 1)
        s += n_or + 1
    n = 256 * B
    buf = np.zeros((n, 16), dtype=np.uint64)
    rc = lib.mlmcpi_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_uint32(n))
    assert rc == 0
    t = buf[:, ORDER].astype(np.int64)
    if n_or <= 7:  # one plane: no stamp 10
        t[:, 3] = t[:, 2]
    t0 = t[:, 0].min()
    seg = (t[:, 1:] - t[:, :-1]) * 0.01  # us (100 MHz)
    print(f"n_or={n_or} B={B}: {n} workgroups, launch span {(t[:, -1].max() - t0) * 0.01:.1f} us")
    for k, name in enumerate(NAMES):
        print(f"  {name:24s} mean {seg[:, k].mean():7.2f}  median {np.median(seg[:, k]):7.2f}  p90 {np.percentile(seg[:, k], 90):7.2f} us")
    life = (t[:, -1] - t[:, 0]) * 0.01
    print(f"  workgroup lifetime       mean {life.mean():7.2f}  median {np.median(life):7.2f} us")
