#!/usr/bin/env python3
"""What would a two-stage pipeline of schwinger_perm_heat_kernel buy?  

A pipeline inside one workgroup (stage A = planes + gathers + image of tile n + 1, stage B = heat bath + write-out of tile n)
is the ENFORCED form of what two independent workgroups on a CU do by chance: A of one beside B of the other.  The stamps of
the instrumentation build (tools/build_variant.sh WORK stamps -DMLMCPI_STAMPS; MLMCPI_LIB_VARIANT=stamps) carry the CU a
workgroup ran on, so the chance pairings can be sorted by what the neighbour was doing: for every workgroup, the share of
its stage A (stage B) during which the OTHER workgroup of its CU was in its stage A / B / absent, and how long the stage
took.  A least-squares fit  duration = t_alone * f_none + t_A * f_A + t_B * f_B  gives the stage times beside each kind of
neighbour; the pipeline's period is max(A beside B, B beside A), today's time per tile is lifetime / 2.

  python tools/exp_stamps_overlap.py [n_or] [chains]      -> table on stdout, raw stamps in gpurun_out/stamps_overlap.npy"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
lib = abi.load()
SEED = 7
n_or = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
x = ops.lattice_initialise(act, B, SEED, 0)
w = torch.empty_like(x)
s = 0
for _ in range(4):
    x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, n_or, 1, SEED, 0, s, 1)
    s += n_or + 1
torch.cuda.synchronize()
n = 256 * B
buf = np.zeros((n, 16), dtype=np.uint64)
assert lib.mlmcpi_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_uint32(n)) == 0
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/stamps_overlap.npy", buf)

t = buf[:, :11].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) * 0.01                                   # us (100 MHz wall clock)
where = buf[:, 15]
hw, xcc = (where & 0xFFFFFFFF).astype(np.int64), (where >> 32).astype(np.int64) & 0xF
cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)   # XCC | SE | SH | CU
start, a_end, end = t[:, 0], t[:, 4], t[:, 9]         # stage A = start .. image down (stamp 4), stage B = .. write-out (9)
print(f"n_or={n_or} B={B}: {n} workgroups on {len(np.unique(cu))} CUs, launch span {end.max():.1f} us, "
      f"lifetime {np.mean(end - start):.2f} us, stage A {np.mean(a_end - start):.2f}, stage B {np.mean(end - a_end):.2f}")

def overlap(lo, hi, lo2, hi2):
    return np.maximum(0.0, np.minimum(hi, hi2) - np.maximum(lo, lo2))

rows_a, rows_b, dur_a, dur_b = [], [], [], []
life_by_offset = []
for c in np.unique(cu):
    idx = np.nonzero(cu == c)[0]
    idx = idx[np.argsort(start[idx])]
    for i in idx:
        others = idx[(idx != i) & (start[idx] < end[i]) & (end[idx] > start[i])]
        for (lo, hi, rows, dur) in ((start[i], a_end[i], rows_a, dur_a), (a_end[i], end[i], rows_b, dur_b)):
            d = hi - lo
            fa = sum(overlap(lo, hi, start[o], a_end[o]) for o in others) / d
            fb = sum(overlap(lo, hi, a_end[o], end[o]) for o in others) / d
            rows.append((max(0.0, 1.0 - fa - fb), fa, fb))
            dur.append(d)
        # where in ITS life the neighbour was when this workgroup started (the neighbour that was running then)
        for o in others:
            if start[o] <= start[i] < end[o]:
                life_by_offset.append(((start[i] - start[o]) / (end[o] - start[o]), end[i] - start[i]))
                break
for name, rows, dur in (("stage A (planes, gathers, image)", rows_a, dur_a), ("stage B (heat bath, write-out)", rows_b, dur_b)):
    X, y = np.array(rows), np.array(dur)
    # duration = sum over kinds of (time spent beside that kind); rate model: 1 = d * (f_none / t_none + f_A / t_A + f_B / t_B)
    coef, *_ = np.linalg.lstsq(X, 1.0 / y, rcond=None)
    with np.errstate(divide="ignore"):
        tt = 1.0 / coef
    print(f"  {name}: mean {y.mean():.2f} us; neighbour absent / in A / in B for {X[:, 0].mean():.2f} / {X[:, 1].mean():.2f} / {X[:, 2].mean():.2f} of it")
    print(f"      fitted duration with the CU to itself {tt[0]:.2f}, beside a stage A throughout {tt[1]:.2f}, beside a stage B throughout {tt[2]:.2f} us")
    for lo in (0.0, 0.25, 0.5, 0.75):
        m = (X[:, 2] >= lo) & (X[:, 2] < lo + 0.25 + (lo == 0.75))
        if m.sum():
            print(f"      share beside a stage B in [{lo:.2f}, {lo + 0.25:.2f}): {m.sum():5d} workgroups, mean {y[m].mean():.2f} us")
lo_ = np.array(life_by_offset)
if len(lo_):
    print("  lifetime by where the neighbour was in its own life when the workgroup started:")
    for k in range(10):
        m = (lo_[:, 0] >= k / 10) & (lo_[:, 0] < (k + 1) / 10)
        if m.sum():
            print(f"      offset {k / 10:.1f}-{(k + 1) / 10:.1f}: {m.sum():5d} workgroups, lifetime mean {lo_[m, 1].mean():.2f}  median {np.median(lo_[m, 1]):.2f} us")
