"""Experiment: 3-level hierarchical Schwinger chain (16x16 -> 8x16 -> 8x8), many chains, vs the exact plaquette."""
import math, sys, torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
sys.path.insert(0, "tests")
from test_gpu_statistics import chain_mean_and_error
beta, B, SEED = 2.0, 256, 77
a0 = abi.lattice_action(4, 16, 16, beta=beta)
a1 = abi.lattice_action(4, 8, 16, beta=beta / 2)
a2 = abi.lattice_action(4, 8, 8, beta=beta / 4)
s0 = ops.LatticeTwoLevelStep(a0, a1, B, seed=SEED + 1)
s1 = ops.LatticeTwoLevelStep(a1, a2, B, seed=SEED + 2)
th = ops.lattice_initialise(a0, B, SEED)
s0.set_state(th)
scr = None
plaq = []
acc0 = acc1 = 0.0
mode = sys.argv[1] if len(sys.argv) > 1 else "batch"
for k in range(1500):
    mid = ops.lattice_copy_from_fine(a0, 2, 1, s0.theta)
    co = ops.lattice_copy_from_fine(a1, 1, 2, mid)
    if scr is None:
        scr = torch.empty_like(co)
    ops.lattice_sweep_draw(a2, co, scr, 1, 1, SEED, 0, 2 * k)
    s1.set_state(mid)
    f1 = s1.draw(co)
    old = s0.theta.clone()
    f0 = s0.draw(s1.theta)
    if mode == "break":   # hierarchicalsampler.cc:62-78: a rejection on a coarser level ends the draw
        s0.theta.copy_(torch.where(f1[:, None] != 0, s0.theta, old))
    acc1 += float(f1.double().mean()); acc0 += float(f0.double().mean())
    if k >= 300:
        plaq.append(ops.qoi_avg_plaquette(s0.theta, 16, 16))
m, e = chain_mean_and_error(torch.stack(plaq))
print(f"plaquette {m:.6f} +- {e:.6f} (exact 0.697775); p_acc level1 {acc1/1500:.3f} level0 {acc0/1500:.3f}")
