"""Experiment: phase timestamps inside schwinger_or_block_kernel (temporary trace hooks)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mlmcpathintegral_amd import abi, ops
lib = abi.load()
seed = 2481317
size, B = 1024, 32
act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)
x = ops.lattice_initialise(act, B, seed, 0)
s = torch.empty_like(x)
abi.set_option("MLMCPI_OR_KERNEL", "block")
nwg = 256 * B
for K in (1, 5):
    for _ in range(3): x, s = ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    tr = torch.zeros((nwg, 6), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    assert lib.mlmcpi_debug_or_trace(C.c_void_p(tr.data_ptr())) == 0
    x, s = ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    torch.cuda.synchronize()
    assert lib.mlmcpi_debug_or_trace(C.c_void_p(0)) == 0
    t = tr.cpu().numpy()
    t0 = t[:, 0].min()
    us = (t[:, :5] - t0) / 100.0   # 100 MHz
    us[:, 4] = us[:, 3]
    load, comp, st_issue, st_done = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 4] - us[:, 3]
    print(f"K={K}: kernel span {us[:,4].max():.1f} us; per WG: load {load.mean():.2f} (p10 {np.percentile(load,10):.2f} p90 {np.percentile(load,90):.2f})  compute {comp.mean():.2f} (p10 {np.percentile(comp,10):.2f}, p90 {np.percentile(comp,90):.2f})  store issue {st_issue.mean():.2f}  store drain {st_done.mean():.2f}  total {(us[:,4]-us[:,0]).mean():.2f}")
    # concurrency over time
    T = us[:, 4].max()
    grid = np.linspace(0, T, 41)[1:-1]
    line = []
    for g in grid[::4]:
        inl = ((us[:, 0] <= g) & (g < us[:, 1])).sum(); inc = ((us[:, 1] <= g) & (g < us[:, 2])).sum(); ins = ((us[:, 2] <= g) & (g < us[:, 4])).sum()
        line.append(f"t={g:.0f}us L{inl} C{inc} S{ins}")
    print("   ", " | ".join(line))
    # per-CU gaps: group by (xcc, hw id cu/se bits)
    hw = t[:, 5]
    key = (hw >> 32) * 100000 + ((hw & 0xffffffff) >> 8 & 0xf) * 100 + ((hw & 0xffffffff) >> 13 & 0x7) * 1000   # cu_id bits 11:8, se_id 15:13
    gaps = []
    for k in np.unique(key):
        idx = np.where(key == k)[0]
        o = idx[np.argsort(us[idx, 0])]
        # with 2 slots per CU: time between a WG end and the next WG start on this CU
        ends = np.sort(us[o, 4]); starts = np.sort(us[o, 0])
        if len(o) > 4: gaps.append((starts[2:] - ends[:-2]).mean())
    print(f"    CUs seen {len(np.unique(key))}, WGs per CU {nwg/len(np.unique(key)):.1f}, mean (start of n+2 - end of n) {np.mean(gaps):.2f} us")
abi.set_option("MLMCPI_OR_KERNEL", "")
