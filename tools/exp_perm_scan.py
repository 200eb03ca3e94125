#!/usr/bin/env python3
"""One launch of K overrelaxation sweeps, Schwinger 1024 x 1024 x 32: closed form (schwinger_perm_kernel) against the
register-block kernel (schwinger_or_block_kernel<K>, K <= 6), and K + heat bath + QoI fused (perm_heat / or_heat<K>, K <= 5)."""
import sys, time
import torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
abi.load()
SEED = 7
act = abi.lattice_action(abi.SCHWINGER, 1024, 1024, beta=1.0)
B = 32
x = ops.lattice_initialise(act, B, SEED, 0)
w = torch.empty_like(x)
def t_or(K, kern, reps=20):
    global x, w
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    for _ in range(3):
        x, w = ops.lattice_sweep_draw_pingpong(act, x, w, K, 0, SEED, 0, 0, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        x, w = ops.lattice_sweep_draw_pingpong(act, x, w, K, 0, SEED, 0, 0, K)
    torch.cuda.synchronize(); abi.set_option("MLMCPI_OR_KERNEL", "")
    return (time.perf_counter() - t0) / reps * 1e3
def t_fused(K, kern, reps=20):
    global x, w
    abi.set_option("MLMCPI_OR_KERNEL", kern)
    for _ in range(3):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, K, 1, SEED, 0, 0, 1, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        x, w, q = ops.lattice_sweep_draw_qoi(act, x, w, x, K, 1, SEED, 0, 0, 1, K)
    torch.cuda.synchronize(); abi.set_option("MLMCPI_OR_KERNEL", "")
    return (time.perf_counter() - t0) / reps * 1e3
for K in range(1, 11):
    a = t_or(K, "")
    b = t_or(K, "block") if K <= 6 else float("nan")
    c = t_fused(K, "")
    d = t_fused(K, "block") if K <= 5 else float("nan")
    print(f"K={K:2d}  OR only: perm {a:.4f} block {b:.4f} ms   K + HB + QoI: perm {c:.4f} block {d:.4f} ms", flush=True)
