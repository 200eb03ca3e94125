#!/usr/bin/env python3
"""Phase timeline of schwinger_or_heat_kernel<5> from in-kernel wall-clock stamps (instrumentation build:
make -C mlmcpathintegral_amd/csrc EXTRA=-DMLMCPI_STAMPS).  Schwinger 1024^2, B chains, draws of 10 OR + 1 HB + QoI."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
lib = abi.load()
SEED = 7
NAMES = ["issue loads", "loads land + publish", "5 OR sweeps", "image down", "HB mu=0 even", "HB mu=0 odd", "HB mu=1 even", "HB mu=1 odd", "write-out + QoI"]
for B in (32, 1):
    act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
    x = ops.lattice_initialise(act, B, SEED, 0)
    w = torch.empty_like(x)
    s = 0
    for _ in range(4):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, 10, 1, SEED, 0, s, 1)
        s += 11
    n = 256 * B
    buf = np.zeros((n, 16), dtype=np.uint64)
    rc = lib.mlmcpi_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_uint32(n))
    assert rc == 0
    t = buf[:, :10].astype(np.int64)
    t0 = t[:, 0].min()
    seg = (t[:, 1:] - t[:, :-1]) * 0.01  # us (100 MHz)
    print(f"B={B}: {n} workgroups, launch span {(t[:, 9].max() - t0) * 0.01:.1f} us")
    for k, name in enumerate(NAMES):
        print(f"  {name:24s} mean {seg[:, k].mean():7.2f}  median {np.median(seg[:, k]):7.2f}  p90 {np.percentile(seg[:, k], 90):7.2f} us")
    life = (t[:, 9] - t[:, 0]) * 0.01
    print(f"  workgroup lifetime       mean {life.mean():7.2f}  median {np.median(life):7.2f} us")
    where = buf[:, 15]
    key = ((where >> 32) << 16) | (where & 0xFF00)
    cus = np.unique(key)
    print(f"  {len(cus)} distinct (XCD, SE/SH/CU) ids; workgroups per id: min {min((key == c).sum() for c in cus)} max {max((key == c).sum() for c in cus)}")
    # per CU: how the two resident workgroups overlap.  At 0.1 us resolution: fraction of the launch with 0 / 1 / 2 workgroups
    # in a heat-bath phase (stamps 4..8) and in the load / overrelaxation part (0..4)
    step = 10  # ticks = 0.1 us
    span = int((t[:, 9].max() - t0) // step) + 1
    hb_both = hb_one = hb_none = 0.0
    for c in cus[:64]:
        m = key == c
        hb = np.zeros(span, dtype=np.int32)
        res = np.zeros(span, dtype=np.int32)
        for a, b4, b8, e in zip(t[m, 0], t[m, 4], t[m, 8], t[m, 9]):
            hb[(b4 - t0) // step:(b8 - t0) // step] += 1
            res[(a - t0) // step:(e - t0) // step] += 1
        hb_both += (hb >= 2).mean()
        hb_one += (hb == 1).mean()
        hb_none += (hb == 0).mean()
    k = min(64, len(cus))
    print(f"  per CU, share of the launch span with 2+ / 1 / 0 workgroups inside the heat-bath phases: {hb_both / k:.2f} / {hb_one / k:.2f} / {hb_none / k:.2f}")
