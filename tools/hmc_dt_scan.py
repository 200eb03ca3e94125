"""Acceptance rate of the fused HMC vs step size at the BASELINE sizes (to pick bench defaults)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mlmcpathintegral_amd import abi, ops
for name, kind, M, m0 in (("rotor", abi.ROTOR, 65536, 0.25), ("quartic", abi.QUARTIC, 32768, 1.0)):
    act = abi.path_action(kind, M, M / 8.0, m0, 1.0, 1.0, 1.0)
    for dt in (0.1, 0.07, 0.05, 0.04, 0.03, 0.02):
        B = 64
        x = ops.path_initialise(act, B, 1)
        hmc = ops.PathHMC(act, B, 100, dt, seed=1)
        for k in range(30):
            hmc.draw(x, count_stats=k >= 10)
        print(name, "M", M, "dt", dt, "p_accept %.3f" % (float(hmc.n_accepted.double().mean()) / hmc.n_total))
