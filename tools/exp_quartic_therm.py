import sys, torch
sys.path.insert(0, ".")
from mlmcpathintegral_amd import abi, ops
M0, T, B = 32768, 1024.0, 64
fine = abi.path_action(abi.QUARTIC, M0, T, 1.0, 1.0, 1.0, 1.0)
x = ops.path_initialise(fine, B, 6)
hmc = ops.PathHMC(fine, B, 100, 0.02, seed=6)
for k in range(60):
    hmc.dt = 0.02 * (0.2 if k < 16 else 0.5 if k < 24 else 1.0)
    acc = hmc.draw(x, count_stats=False)
    en = hmc.energies
    dH = (en[:, 2] - en[:, 0]) + (en[:, 3] - en[:, 1])
    q = ops.qoi_xsquared(x)
    if k < 30 or k % 10 == 0:
        print(k, "dt %.4f" % hmc.dt, "acc %.2f" % float(acc.double().mean()), "dH med %.3g max %.3g" % (float(dH.median()), float(dH.max())),
              "x2 min %.4f mean %.4f" % (float(q.min()), float(q.mean())))
