"""Check / timing: gff_or_block_kernel (4 x 4 register blocks, 64 x 64 tiles) against gff_or_patch_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
abi.load()
seed = 2481317
def sweeps(act, x, K, kernel):
    abi.set_option("MLMCPI_OR_KERNEL", kernel)
    y, s = x.clone(), torch.empty_like(x)
    y, s = ops.lattice_sweep_draw_pingpong(act, y, s, K, 0, seed, 0, 0, K)
    return y
bad = 0
for (Mt, Mx) in ((64, 64), (128, 128), (192, 192), (512, 512)):
    act = abi.lattice_action(abi.GFF, Mt, Mx, mass=10.0)
    x = ops.lattice_initialise(act, 3, seed, 0)
    for K in range(1, 7):
        a, b = sweeps(act, x, K, "block"), sweeps(act, x, min(K, 4), "patch")
        if K > 4: b = sweeps(act, b, K - 4, "patch")
        ok = torch.equal(a, b); bad += not ok
        print(f"gff {Mt}x{Mx} K={K}: block == patch: {ok}" + ("" if ok else f"  max diff {(a-b).abs().max().item():.3e}"))
print("MISMATCHES" if bad else "all equal")
size, B = 512, 1024
act = abi.lattice_action(abi.GFF, size, size, mass=10.0)
x = ops.lattice_initialise(act, B, seed, 0)
s = torch.empty_like(x)
def timeit(K, n=10):
    for _ in range(2): ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.lattice_sweep_draw_pingpong(act, x, s, K, 0, seed, 0, 0, K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    for K in range(1, 7):
        out = []
        for kernel in ("patch", "block"):
            if kernel == "patch" and K > 4: out.append("   -  "); continue
            abi.set_option("MLMCPI_OR_KERNEL", kernel)
            out.append(f"{timeit(K):.4f}")
        print(f"rep {rep} K={K}: patch {out[0]}  block {out[1]} ms/launch")
abi.set_option("MLMCPI_OR_KERNEL", "")
sys.exit(1 if bad else 0)
