"""Experiment / check: the 4 x 4 register-block overrelaxation kernel (64 x 64 tiles) against the 2 x 2 patch kernel
(64 x 32 tiles): bit-equality of K fused sweeps on several lattice shapes, then launch times at 1024 x 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
abi.load()
seed = 2481317

def sweeps(act, x, K, kernel, total=None):
    abi.set_option("MLMCPI_OR_KERNEL", kernel)
    y, s = x.clone(), torch.empty_like(x)
    y, s = ops.lattice_sweep_draw_pingpong(act, y, s, total or K, 0, seed, 0, 0, K)
    return y

bad = 0
for (Mt, Mx) in ((64, 64), (128, 64), (64, 128), (192, 128), (1024, 1024)):
    act = abi.lattice_action(abi.SCHWINGER, Mt, Mx, beta=1.0)
    B = 3 if Mt < 1024 else 2
    x = ops.lattice_initialise(act, B, seed, 0)
    for K in range(1, 7):
        a, b = sweeps(act, x, K, "block"), sweeps(act, x, K, "patch" if K <= 4 else "lds")
        ok = torch.equal(a, b)
        bad += not ok
        print(f"{Mt}x{Mx} K={K}: block == reference kernel: {ok}" + ("" if ok else f"  max diff {(a-b).abs().max().item():.3e}"))
print("MISMATCHES" if bad else "all equal")

size, B = 1024, 32
act = abi.lattice_action(abi.SCHWINGER, size, size, beta=1.0)
x = ops.lattice_initialise(act, B, seed, 0)
s = torch.empty_like(x)
def timeit(K, n=20):
    for _ in range(3): ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):  # the variants alternate inside one process: clocks and box are the same for both
    for K in range(1, 7):
        out = []
        for kernel in ("patch", "block"):
            if kernel == "patch" and K > 4: out.append("   -  "); continue
            abi.set_option("MLMCPI_OR_KERNEL", kernel)
            out.append(f"{timeit(K):.4f}")
        print(f"rep {rep} K={K}: 2x2 patch {out[0]}  4x4 blocks {out[1]} ms/launch")
abi.set_option("MLMCPI_OR_KERNEL", "")
sys.exit(1 if bad else 0)
